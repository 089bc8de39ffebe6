#!/bin/bash
# Round profile of the headline command (run on the GPU box through gpurun; writes under gpurun_out/prof_$TAG):
#   kernel stats (rocprofv3 --kernel-trace --stats) and three separate --pmc passes (FETCH_SIZE / WRITE_SIZE / MFMA
#   counters), as MI355X_MICROARCH.md prescribes (counters never combined with trace domains beyond --kernel-trace).
# usage: tools/profile_round.sh TAG [extra bench.py args, e.g. --precision bf16]   (default mode: exact fp32)
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
# (step counts are multiples of 32: every k_ddpg_chain launch then holds 32 updates and per-launch figures divide evenly)
BENCH="python bench.py --steps 3200 --warmup 320 --pre-warm 3200 --no-cpu-baseline --learners 0 --no-configs --profile-steps 32 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $BENCH > /dev/null 2>&1
BENCH="python bench.py --steps 640 --warmup 96 --no-cpu-baseline --learners 0 --no-configs --profile-steps 32 --pre-warm 192 $*"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- $BENCH > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- $BENCH > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -o m -- $BENCH > /dev/null 2>&1
head -1 $OUT/stats/*kernel_stats.csv > $OUT/kernel_stats.csv
grep -E "oprl|k_replay" $OUT/stats/*kernel_stats.csv >> $OUT/kernel_stats.csv
OPRL_UPL=32 python tools/pmc_summary.py $OUT/fetch $OUT/write $OUT/pmc_traffic.json $OUT/mfma $OUT/kernel_stats.csv > /dev/null
cat $OUT/pmc_traffic.json
rm -rf $OUT/stats $OUT/fetch $OUT/write $OUT/mfma       # (raw traces: scratch)
