"""Summarise rocprofv3 --pmc passes (one counter group per pass, as MI355X_MICROARCH.md prescribes)
into per-kernel HBM bytes per launch and, with a fourth argument, matrix-core utilisation.
``python tools/pmc_summary.py <fetch_dir> <write_dir> <out.json> [<mfma_dir>]``
FETCH_SIZE is doubled (gfx950 reports half of a wide coalesced read); rocprofv3 reports KB.  The MFMA pass holds
SQ_INSTS_VALU_MFMA_MOPS_F32 / _BF16 (x 512 = flops), SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over the chip's 1024
SIMDs), SQ_BUSY_CYCLES and GRBM_GUI_ACTIVE; with a fifth argument (the --stats run's kernel_stats.csv) mfma_util =
MFMA busy SIMD-cycles / (un-profiled average kernel duration x 2.4 GHz x 1024 SIMDs)."""
import collections
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        return {}
    f = fs[0]
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
# k_ddpg_chain runs several updates per launch (OPRL_UPL of them in the profiled command, tools/profile_round.sh)
import os
_upl = float(os.environ.get("OPRL_UPL", "0") or 0)
if _upl > 0 and "k_ddpg_chain" in out:
    out["k_ddpg_chain"]["updates_per_launch"] = _upl
    out["k_ddpg_chain"]["hbm_bytes_per_update"] = int(out["k_ddpg_chain"]["hbm_bytes_per_launch"] / _upl)


def stats_ns(path):
    """kernel short name -> average duration (ns) from a rocprofv3 --stats kernel_stats.csv"""
    out = {}
    for r in csv.DictReader(open(path)):
        k = r["Name"].replace("void ", "").replace("oprl::", "").split("<")[0].split("(")[0]
        out[k] = float(r["AverageNs"])
    return out


CLOCK_GHZ = 2.4      # MI355X_MICROARCH.md max clock; in-kernel stamps read 2.31-2.40 GHz for these kernels
if len(sys.argv) > 4:
    c = {n: per_kernel(sys.argv[4], n) for n in ("SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_VALU_MFMA_MOPS_BF16",
                                                 "SQ_INSTS_VALU_MFMA_MOPS_F16",
                                                 "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE")}
    for k in out:
        if k not in c["GRBM_GUI_ACTIVE"]:
            continue
        gui = c["GRBM_GUI_ACTIVE"][k][0]
        busy = c["SQ_VALU_MFMA_BUSY_CYCLES"].get(k, (0.0, 0))[0]
        ns = stats_ns(sys.argv[5]).get(k) if len(sys.argv) > 5 else None
        out[k].update({
            "mfma_flops_f32_per_launch": int(512 * c["SQ_INSTS_VALU_MFMA_MOPS_F32"].get(k, (0.0, 0))[0]),
            "mfma_flops_bf16_per_launch": int(512 * c["SQ_INSTS_VALU_MFMA_MOPS_BF16"].get(k, (0.0, 0))[0]),
            "mfma_flops_f16_per_launch": int(512 * c["SQ_INSTS_VALU_MFMA_MOPS_F16"].get(k, (0.0, 0))[0]),
            "mfma_busy_cycles_per_launch": round(busy, 1), "gui_active_cycles_per_launch": round(gui, 1),
            "sq_busy_cycles_per_launch": round(c["SQ_BUSY_CYCLES"].get(k, (0.0, 0))[0], 1),
            "avg_duration_ns_unprofiled": ns,
            # busy SIMD-cycles / (kernel duration x clock x 1024 SIMDs); the duration is the un-profiled run's
            # average (rocprofv3 --stats: back-to-back launches, i.e. it includes the ~2 us launch gap)
            "mfma_util": round(busy / (ns * CLOCK_GHZ * 1024.0), 5) if ns else None})
out["_note"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes with --kernel-trace only; "
                "hbm bytes = (2*FETCH_SIZE + WRITE_SIZE) KB: FETCH_SIZE doubled per MI355X_MICROARCH.md "
                "(gfx950 reports half of a wide coalesced read); mfma_util = SQ_VALU_MFMA_BUSY_CYCLES (SIMD-cycles, = 32 per v_mfma_f32_16x16x4_f32) / (average kernel duration of the "
                "un-profiled --stats run x 2.4 GHz x 1024 SIMDs) "
                "from a third pass (SQ_INSTS_VALU_MFMA_MOPS_* x 512 = matrix flops executed, padding included)")
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
