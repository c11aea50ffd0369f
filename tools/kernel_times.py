"""Developer tool: per-kernel time of one bench.py run (rocprofv3 kernel trace), as a small table.
usage: python tools/kernel_times.py <out_dir> -- <bench.py args>     (env knobs pass through)"""
import glob
import os
import sqlite3
import subprocess
import sys

out = sys.argv[1]
args = sys.argv[3:] if len(sys.argv) > 2 and sys.argv[2] == "--" else sys.argv[2:]
os.makedirs(out, exist_ok=True)
cmd = ["rocprofv3", "--kernel-trace", "-d", out, "--", sys.executable, "bench.py", *args]
with open(os.path.join(out, "run.log"), "w") as log:
    subprocess.run(cmd, stdout=log, stderr=subprocess.STDOUT, check=False, env={**os.environ, "TMPDIR": "/tmp"})
db = sorted(glob.glob(os.path.join(out, "**", "*_results.db"), recursive=True))[-1]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "info_kernel_symbol" in t][0]
q = (f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3 from {kd} d "
     f"join {ks} s on d.kernel_id=s.id group by 1 order by 3 desc")
print("kernel,calls,avg_us,min_us")
for name, n, avg, mn in c.execute(q):
    if "fqtk" in name:
        print(f"{name},{n},{avg:.1f},{mn:.1f}")
