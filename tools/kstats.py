"""Per-kernel launches and average duration of `tools/quick_cfg.py CONFIG spec` under rocprofv3 (needs a GPU):
`python tools/kstats.py CONFIG spec` -> one line per kernel (name, calls, average us), sorted by total time."""
import csv
import glob
import os
import shutil
import subprocess
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
out = "/tmp/kstats_prof"
shutil.rmtree(out, ignore_errors=True)
env = dict(os.environ, TMPDIR="/tmp")
subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", out, "-o", "s", "--",
                sys.executable, str(ROOT / "tools" / "quick_cfg.py")] + sys.argv[1:], cwd="/tmp", env=env,
               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
rows = []
for f in glob.glob(out + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "oprl" in r["Name"] or "k_replay" in r["Name"]:
            rows.append((float(r["TotalDurationNs"]), r["Name"], int(r["Calls"]), float(r["AverageNs"])))
for tot, name, calls, avg in sorted(rows, reverse=True):
    print(f"{name[:96]:96s} {calls:6d} x {avg / 1e3:8.2f} us")
