#!/bin/bash
# Round 4, fourth GPU call: same-box A/B of the round-3 tree against the current one (default arithmetic and screen off)
# after the per-root thresholds went to integer exponent compares and the cell phase keeps its items in registers.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4d
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R && timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
cd /tmp
for i in 1 2; do
(cd $R/ab_r3 && timeout 300 python bench.py --no-cpu --no-host-leg --steps 20 --warmup 5 2>/dev/null > $O/bench_r3_$i.json)
timeout 300 python $R/bench.py --no-cpu --no-host-leg --steps 20 --warmup 5 --vote-arith 2 --no-false-hint-leg 2>/dev/null > $O/bench_arith2_$i.json
timeout 300 python $R/bench.py --no-cpu --no-host-leg --steps 20 --warmup 5 --no-false-hint-leg 2>/dev/null > $O/bench_arith1_$i.json
done
timeout 300 python $R/bench.py --no-cpu --no-host-leg --steps 5 --config C3 --frames 65536 2>$O/bench_C3.err > $O/bench_C3.json
cd $R
timeout 900 python tests/soak_votes.py 32768 C2 gpurun_out/r4d/soak_votes_C2 > $O/soak_votes_C2.log 2>&1; echo "rc $?" >> $O/soak_votes_C2.log
timeout 900 python tests/soak_votes.py 2048 C3 gpurun_out/r4d/soak_votes_C3 > $O/soak_votes_C3.log 2>&1; echo "rc $?" >> $O/soak_votes_C3.log
ls $O
