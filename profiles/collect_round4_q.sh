#!/bin/bash
# Round 4, seventeenth GPU call: ONE block per CU, larger shares; the side scan's share / resident blocks once more — with the shorter voting launch the
# side scan (3 blocks per CU, 30 % of a sub-batch) no longer finishes inside blob window + vote (call m: both kernels
# faster, the step 0.25 ms slower).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4q
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {  # name, args
  local name=$1; shift
  timeout 200 python $R/bench.py --no-cpu --no-host-leg --steps 15 --warmup 5 --no-false-hint-leg "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernel_ms']
print('$name', round(d['ms_per_step'], 3), round(d['ms_per_step_median'], 3), 'vote_in_region', d['roofline'].get('avg_launch_ms'), 'blobs', round(k['blobs'], 3), 'vote', round(k['vote'], 3), 'tail', round(k['tail'], 3))" >> $O/sweep.log 2>&1
}
run 31_1 --scan-split-pct 31 --side-scan-blocks 1
run 36_1 --scan-split-pct 36 --side-scan-blocks 1
run 42_1 --scan-split-pct 42 --side-scan-blocks 1
cat $O/sweep.log
