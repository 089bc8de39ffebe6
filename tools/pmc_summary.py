"""Summarise rocprofv3 --pmc passes (one counter per pass, as MI355X_MICROARCH.md prescribes)
into per-kernel HBM bytes per launch.  ``python tools/pmc_summary.py <fetch_dir> <write_dir> <out.json>``
FETCH_SIZE is doubled (gfx950 reports half of a wide coalesced read); rocprofv3 reports KB."""
import collections
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    tot, n = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        k = k.replace("void ", "").replace("oprl::", "").split("<")[0].split("(")[0]
        tot[k] += float(r["Counter_Value"])
        n[k] += 1
    return {k: (tot[k] / n[k], n[k]) for k in tot}


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in fetch:
    if not k.startswith("k_"):
        continue
    f_kb, n = fetch[k]
    w_kb = write.get(k, (0.0, 0))[0]
    out[k] = {"FETCH_SIZE_KB_per_launch_raw": round(f_kb, 1), "WRITE_SIZE_KB_per_launch": round(w_kb, 1),
              "launches": n, "hbm_bytes_per_launch": int((2 * f_kb + w_kb) * 1024)}
out["_note"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes with --kernel-trace only; "
                "hbm bytes = (2*FETCH_SIZE + WRITE_SIZE) KB: FETCH_SIZE doubled per MI355X_MICROARCH.md "
                "(gfx950 reports half of a wide coalesced read)")
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
