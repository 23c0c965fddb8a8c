"""Per-kernel totals of a rocprofv3 kernel trace, grouped by kernel name and grid:
    python tools/trace_summary.py <results.db | *_kernel_trace.csv> [rows]
(a rocpd database, rocprofv3's default output, or the CSV of --output-format csv)."""
import collections
import csv
import re
import sqlite3
import sys

path = sys.argv[1]
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 30
groups = collections.defaultdict(list)
if path.endswith(".csv"):
    for r in csv.DictReader(open(path)):
        groups[(r["Kernel_Name"], int(r["Grid_Size_X"]), int(r["Grid_Size_Z"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
else:
    db = sqlite3.connect(path)
    for n, gx, gz, s, e in db.execute("select name, grid_x, grid_z, start, end from kernels"):
        groups[(n, gx, gz)].append(e - s)
tot = sum(sum(v) for v in groups.values()) / 1e6
print(f"total kernel time {tot:.2f} ms")
for (n, gx, gz), v in sorted(groups.items(), key=lambda kv: -sum(kv[1]))[:rows]:
    n = re.sub(r"btvae_gemm::|\(anonymous namespace\)::|void ", "", n)
    print(f"{sum(v) / 1e6:9.2f} ms {len(v):5d} x {sum(v) / len(v) / 1e3:9.1f} us  grid {gx:8d} z{gz:4d}  {n[:90]}")
