#!/bin/bash
# Profile of one BASELINE config (tools/bench_algos.py) — kernel stats and FETCH / WRITE counter passes, separately
# (MI355X_MICROARCH.md: counters never combined with trace domains beyond --kernel-trace).
# usage: tools/profile_config.sh TAG "<case substring>" <precision> [updates]
TAG=$1; CASE=$2; PREC=$3; N=${4:-600}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/cfg_$TAG
CMD="python tools/bench_algos.py $N"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $CMD "$CASE" $PREC > $OUT.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python tools/bench_algos.py 200 "$CASE" $PREC > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python tools/bench_algos.py 200 "$CASE" $PREC > /dev/null 2>&1
head -1 $OUT/stats/*kernel_stats.csv > $OUT/kernel_stats.csv
grep -E "oprl|k_replay" $OUT/stats/*kernel_stats.csv >> $OUT/kernel_stats.csv
python tools/pmc_summary.py $OUT/fetch $OUT/write $OUT/pmc_traffic.json > /dev/null
grep -E "update\(\)" $OUT.log
rm -rf $OUT/stats $OUT/fetch $OUT/write       # (raw traces: scratch)
