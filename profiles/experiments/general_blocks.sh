#!/bin/bash
# Experiment (round 5): waves of the general blob tier in flight — the salt-noise leg and the clean headline (whose
# every sub-batch launches the tier empty) at 32 / 1024 / 2048 / 4096 blocks.
O=gpurun_out/r5d; mkdir -p $O
Q="--no-cpu --no-host-leg --no-false-hint-leg --headline-only"
for n in 1024 4096 32 2048 1024 4096; do
  python bench.py $Q --steps 15 --warmup 4 --opt k1b_general_blocks=$n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', $n, round(d['ms_per_step'],3), round(d['ms_per_step_median'],3), [round(x['blobs'],3) for x in d['kernel_ms']['per_sub_batch'][1:]])" >> $O/out.txt
done
for n in 1024 2048 4096 8192; do
  python bench.py $Q --clutter salt --frames 32768 --steps 5 --warmup 2 --opt k1b_general_blocks=$n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('salt', $n, round(d['value']), round(d['ms_per_step'],3), round(d['kernel_ms']['blobs'],3))" >> $O/out.txt
  python bench.py $Q --clutter patch --frames 32768 --steps 5 --warmup 2 --opt k1b_general_blocks=$n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('patch', $n, round(d['value']), round(d['ms_per_step'],3), round(d['kernel_ms']['blobs'],3))" >> $O/out.txt
done
cat $O/out.txt
