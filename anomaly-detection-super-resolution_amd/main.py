"""CLI entry with the reference's flags (src/main.py:390-474).  Inference / evaluation of existing
checkpoints runs on the HIP engine; the training step (fwd + L1 + backward + Adam, data-parallel over
RCCL) is the next row of the build (DESIGN.md "What comes next") and exits with a clear message."""
from __future__ import annotations

import sys

from .options import parse_train_args


def main(argv=None):
    args = parse_train_args(argv)
    print(f"Model: {args.model_type}\nDataset: {args.dataset}\nClass: {args.classe}\n"
          f"Resolution: {args.resolution}\nScale: {args.scale}")
    if args.device == 'cpu':
        raise SystemExit("--device cpu is the reference's own path; this build has no CPU fallback")
    if not args.test_only:
        raise SystemExit("training on the HIP engine is not built yet (backward kernels + fused Adam + RCCL "
                         "all-reduce are the next rows, see DESIGN.md); use --test-only or srad_amd.evaluate")
    from . import evaluate
    return evaluate.main([a for a in (argv if argv is not None else sys.argv[1:]) if a != '--test-only'])


if __name__ == "__main__":
    main()
