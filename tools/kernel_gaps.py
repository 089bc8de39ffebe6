"""Debug tool: per-kernel durations and the idle gap before each kernel, from a
rocprofv3 --kernel-trace CSV (``python tools/kernel_gaps.py <dir>``)."""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "phase1" in r["Kernel_Name"]]
i0 = idx[len(idx) // 2]
prev_end = None
for r in rows[i0:i0 + 9]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(r["Kernel_Name"][:48].ljust(48), "dur", e - s, "gap", (s - prev_end) if prev_end else None)
    prev_end = e
d, g = collections.defaultdict(list), collections.defaultdict(list)
pe = None
for r in rows[idx[len(idx) // 4]:idx[-len(idx) // 8]]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    d[r["Kernel_Name"][:34]].append(e - s)
    if pe:
        g[r["Kernel_Name"][:34]].append(s - pe)
    pe = e
for k in d:
    print(k.ljust(36), "n", len(d[k]), "avg dur us %.2f" % (sum(d[k]) / len(d[k]) / 1e3),
          "avg gap before us %.2f" % (sum(g[k]) / max(1, len(g[k])) / 1e3))
