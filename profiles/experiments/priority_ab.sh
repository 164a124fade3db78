#!/bin/bash
# Experiment (round 5): which non-default priority levels for the two side streams?  Interleaved repetitions on one box.
O=gpurun_out/r5l; mkdir -p $O
Q="--no-cpu --no-host-leg --no-false-hint-leg --headline-only --steps 20 --warmup 5"
run() {
  python bench.py $Q "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['timed_region_by_slot']; k=d['kernel_ms']['per_sub_batch'][1:7]; print('$*', round(d['ms_per_step'],3), round(d['ms_per_step_median'],3), 'gaps', round(sum(x['gap_before_ms'] for x in s)/8,3), 'launch', round(sum(x['launch_ms'] for x in s)/8,3), 'tail', round(sum(x['tail'] for x in k)/6,3))" >> $O/out.txt
}
for rep in 1 2 3 4; do
run --opt tail_priority=1 --opt scan_priority=-1
run --opt tail_priority=-1 --opt scan_priority=-1
run --opt tail_priority=-1 --opt scan_priority=1
run --opt tail_priority=1 --opt scan_priority=1
done
cat $O/out.txt
