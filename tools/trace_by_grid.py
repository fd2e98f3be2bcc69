"""Average dispatch duration per (kernel, grid size) from a rocprofv3 --kernel-trace CSV directory (tools only)."""
import csv, glob, sys, collections
rows = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        full = r["Kernel_Name"]
        name = full.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
        g = r.get("Grid_Size") or r.get("Grid_Size_X")
        rows[(name, g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (name, g), v in sorted(rows.items()):
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    v = v[len(v) // 4:]          # drop the warm-up quarter
    print(f"{name:60s} grid {g:>8s}  n={len(v):4d}  avg {sum(v)/len(v):8.2f} us  min {min(v):8.2f}")
