"""Per-kernel totals of a rocprofv3 kernel trace kept as a rocpd database: python tools/trace_summary.py <results.db> [rows]."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 30
tot = db.execute("select sum(end-start)/1e6 from kernels").fetchone()[0]
print(f"total kernel time {tot:.2f} ms")
q = ("select name, grid_x, grid_z, count(*), avg(end-start)/1e3, sum(end-start)/1e6 from kernels "
     "group by name, grid_x, grid_z order by 6 desc limit ?")
for n, gx, gz, k, a, t in db.execute(q, (rows,)):
    n = re.sub(r"btvae_gemm::|\(anonymous namespace\)::|void ", "", n)
    print(f"{t:9.2f} ms {k:5d} x {a:9.1f} us  grid {gx:8d} z{gz:4d}  {n[:90]}")
