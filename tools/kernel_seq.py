"""Debug tool: the launch sequence of ONE update out of a rocprofv3 --kernel-trace CSV, averaged per
position over the middle half of the run (``python tools/kernel_seq.py <dir> <anchor substring>``: an
update starts at each launch whose name contains the anchor, e.g. k_tqc_target)."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
anchor = sys.argv[2]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
idx = idx[len(idx) // 4: -len(idx) // 4]
per = {}
for a, b in zip(idx[:-1], idx[1:]):
    names = tuple(r["Kernel_Name"] for r in rows[a:b])
    acc = per.setdefault(names, [0, [0.0] * (b - a), [0.0] * (b - a)])
    acc[0] += 1
    for k, r in enumerate(rows[a:b]):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        acc[1][k] += e - s
        acc[2][k] += s - int(rows[a + k - 1]["End_Timestamp"])
for names, (n, dur, gap) in sorted(per.items(), key=lambda kv: -kv[1][0])[:2]:
    tot = 0.0
    print(f"--- {n} updates with this sequence of {len(names)} launches")
    for k, nm in enumerate(names):
        d, g = dur[k] / n / 1e3, gap[k] / n / 1e3
        tot += d + g
        print(f"{k:3d} {nm[:70]:70s} dur {d:7.2f}  gap before {g:6.2f}")
    print(f"    period {tot:.1f} us")
