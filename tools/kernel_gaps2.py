"""Durations of, and idle gaps before, the kernels of a rocprofv3 --kernel-trace CSV whose name contains `pat`
(`python tools/kernel_gaps2.py <dir> [pat]`): the middle half of the trace."""
import csv
import glob
import sys
import statistics as st

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_ddpg_chain"
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 4: 3 * len(rows) // 4]
dur, gap, pe = [], [], None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if pat in r["Kernel_Name"]:
        dur.append((e - s) / 1e3)
        if pe is not None:
            gap.append((s - pe) / 1e3)
    pe = e
q = lambda v, p: sorted(v)[int(p * (len(v) - 1))]
print(f"{pat}: n {len(dur)}  duration us: min {min(dur):.2f} median {st.median(dur):.2f} p90 {q(dur, .9):.2f} max {max(dur):.2f};  "
      f"gap before us: min {min(gap):.2f} median {st.median(gap):.2f} p90 {q(gap, .9):.2f} max {max(gap):.2f}")
