#!/bin/bash
# Profile of the packed learner group (tools/probe_group.py 32 <precision>): kernel stats and FETCH / WRITE counter passes,
# separately (MI355X_MICROARCH.md: counters never combined with trace domains beyond --kernel-trace).
# usage: tools/profile_group.sh TAG <precision>
TAG=$1; PREC=${2:-f32}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/grp_$TAG
CMD="python tools/probe_group.py 32 $PREC"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $CMD > $OUT.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- $CMD > /dev/null 2>&1
head -1 $OUT/stats/*kernel_stats.csv > $OUT/kernel_stats.csv
grep -E "oprl|k_replay" $OUT/stats/*kernel_stats.csv >> $OUT/kernel_stats.csv
python tools/pmc_summary.py $OUT/fetch $OUT/write $OUT/pmc_traffic.json > /dev/null
grep -E "group of" $OUT.log
rm -rf $OUT/stats $OUT/fetch $OUT/write       # (raw traces: scratch)
