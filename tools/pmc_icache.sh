#!/bin/bash
# instruction-cache behaviour of the headline launch (one --pmc pass, --kernel-trace only)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_icache
BENCH="python bench.py --steps 600 --warmup 100 --no-cpu-baseline --learners 0 --no-configs --profile-steps 1 --pre-warm 200 $*"
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $OUT/ic -o i -- $BENCH > $OUT.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/sq -o q -- $BENCH >> $OUT.log 2>&1
python - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/prof_icache/ic", "gpurun_out/prof_icache/sq"):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(d, "no counters (see gpurun_out/prof_icache.log)"); continue
    tot, n = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(fs[0])):
        k = (r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])
        tot[k] += float(r["Counter_Value"]); n[k] += 1
    for k in sorted(tot):
        print(k, round(tot[k] / n[k], 1), n[k])
PY
tail -5 $OUT.log
