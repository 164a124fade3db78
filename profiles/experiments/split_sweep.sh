#!/bin/bash
# Experiment (round 5): the side scan's share of a sub-batch, re-swept after the side streams got hardware queues of
# their own (round 4's optimum, 28 %, was found with the scan stream sharing a queue).
O=gpurun_out/r5i; mkdir -p $O
Q="--no-cpu --no-host-leg --no-false-hint-leg --headline-only --steps 15 --warmup 4"
run() {
  python bench.py $Q "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['timed_region_by_slot']; print('$*', round(d['ms_per_step'],3), round(d['ms_per_step_median'],3), 'gaps', round(sum(x['gap_before_ms'] for x in s)/8,3), 'launch', round(sum(x['launch_ms'] for x in s)/8,3))" >> $O/out.txt
}
for rep in 1 2; do
for p in 28 22 25 31 34 19; do run --scan-split-pct $p; done
run --scan-split-pct 34 --side-scan-blocks 2
run --scan-split-pct 40 --side-scan-blocks 2
done
cat $O/out.txt
