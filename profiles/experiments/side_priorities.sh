#!/bin/bash
# Experiment (round 5): stream priority of the tail stream (fix-up + validate + refine) and of the side-scan stream.
O=gpurun_out/r5e; mkdir -p $O
Q="--no-cpu --no-host-leg --no-false-hint-leg --headline-only --steps 15 --warmup 4"
run() {
  python bench.py $Q "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']['per_sub_batch'][1:7]; print('$*', round(d['ms_per_step'],3), round(d['ms_per_step_median'],3), 'blobs', round(sum(x['blobs'] for x in k)/6,3), 'vote', round(sum(x['vote'] for x in k)/6,3), 'tail', round(sum(x['tail'] for x in k)/6,3))" >> $O/out.txt
}
python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('by_slot', d['ms_per_step'], json.dumps(d['roofline']['timed_region_by_slot']))" >> $O/out.txt
for rep in 1 2; do
run
run --opt tail_priority=1
run --opt tail_priority=-1
run --opt scan_priority=1
run --opt scan_priority=-1
run --opt tail_priority=1 --opt scan_priority=1
run --opt tail_priority=-1 --opt scan_priority=1
done
cat $O/out.txt
