for k in 0 2 4; do
  echo "== SRAD_WGRAD_KSPLIT=$k"
  if [ $k != 0 ]; then export SRAD_WGRAD_KSPLIT=$k; fi
  bash $GRAFT_REPO_ROOT/tools/wgrad_trace.sh
done
