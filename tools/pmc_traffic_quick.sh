#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the headline launch, two quick passes (usage: tools/pmc_traffic_quick.sh TAG [bench args])
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
BENCH="python bench.py --steps 600 --warmup 100 --no-cpu-baseline --learners 0 --no-configs --profile-steps 1 --pre-warm 200 $*"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/q_$TAG/f -o f -- $BENCH > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/q_$TAG/w -o w -- $BENCH > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/q_$TAG/f gpurun_out/q_$TAG/w gpurun_out/q_$TAG/t.json | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    if k.startswith('k_ddpg'): print('$TAG', k, 'FETCH raw KB', v['FETCH_SIZE_KB_per_launch_raw'], 'WRITE KB', v['WRITE_SIZE_KB_per_launch'], 'bytes', v['hbm_bytes_per_launch'])"
