#!/bin/bash
# Experiment (round 5): the tail chain of sub-batch s held back until blobs(s + 1) is done (option tail_after_blobs),
# with k2_vote_relost as one-wave blocks; interleaved A/B on one box + the GPU suite on the new default.
O=gpurun_out/r5o; mkdir -p $O
Q="--no-cpu --no-host-leg --no-false-hint-leg --headline-only --steps 20 --warmup 5"
run() {
  python bench.py $Q "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['timed_region_by_slot']; k=d['kernel_ms']['per_sub_batch'][1:7]; print('$*', round(d['ms_per_step'],3), round(d['ms_per_step_median'],3), 'gaps', round(sum(x['gap_before_ms'] for x in s)/8,3), 'launch', round(sum(x['launch_ms'] for x in s)/8,3), 'tail', round(sum(x['tail'] for x in k)/6,3))" >> $O/out.txt
}
for rep in 1 2 3 4; do
run --opt tail_after_blobs=1
run --opt tail_after_blobs=0
done
run --opt tail_after_blobs=1 --opt tail_priority=-1 --opt scan_priority=-1
run --opt tail_after_blobs=1 --opt tail_priority=-1 --opt scan_priority=-1
timeout 500 python -m pytest tests -m gpu -q -x 2>&1 | tail -2 >> $O/out.txt
cat $O/out.txt
