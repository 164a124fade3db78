#!/bin/bash
# Experiment (round 5): what stretches the blob window from 0.31 ms (kernel alone) to 0.5 - 0.56?  The side scan's
# share swept down to nothing, the tail at the lowest priority, both.
O=gpurun_out/r5k; mkdir -p $O
Q="--no-cpu --no-host-leg --no-false-hint-leg --headline-only --steps 15 --warmup 4"
run() {
  python bench.py $Q "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['timed_region_by_slot']; k=d['kernel_ms']['per_sub_batch'][1:7]; print('$*', round(d['ms_per_step'],3), round(d['ms_per_step_median'],3), 'gaps', round(sum(x['gap_before_ms'] for x in s)/8,3), 'launch', round(sum(x['launch_ms'] for x in s)/8,3), 'blobs(profiled)', round(sum(x['blobs'] for x in k)/6,3))" >> $O/out.txt
}
for rep in 1 2; do
run
run --scan-split-pct 0
run --scan-split-pct 10
run --opt tail_priority=-1
run --scan-split-pct 0 --opt tail_priority=-1
run --pipeline-mode 3
done
cat $O/out.txt
