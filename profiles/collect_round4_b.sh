#!/bin/bash
# Round 4, second GPU call: the suspect list buffered in LDS (no global atomics inside the voting loop) — A/B against the
# fast arithmetic alone on one box, the vote tests, the vote soaks again.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4b
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R && timeout 900 python -m pytest tests -m gpu -x -q -k "vote or strict or streaming or golden or estimate_batch_parity or bruteforce" > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
cd /tmp
for i in 1 2; do
timeout 300 python $R/bench.py --no-cpu --no-host-leg --steps 20 --warmup 5 --vote-arith 2 --no-false-hint-leg 2>/dev/null > $O/bench_arith2_$i.json
timeout 300 python $R/bench.py --no-cpu --no-host-leg --steps 20 --warmup 5 --no-false-hint-leg 2>/dev/null > $O/bench_arith1_$i.json
done
timeout 300 python $R/bench.py --no-cpu --no-host-leg --steps 5 --config C3 --frames 65536 2>$O/bench_C3.err > $O/bench_C3.json
timeout 300 python $R/bench.py --no-cpu --no-host-leg --steps 5 --config C3 --frames 65536 --vote-arith 2 2>/dev/null > $O/bench_C3_arith2.json
cd $R
timeout 900 python tests/soak_votes.py 65536 C2 gpurun_out/r4b/soak_votes_C2 > $O/soak_votes_C2.log 2>&1; echo "rc $?" >> $O/soak_votes_C2.log
timeout 900 python tests/soak_votes.py 4096 C3 gpurun_out/r4b/soak_votes_C3 > $O/soak_votes_C3.log 2>&1; echo "rc $?" >> $O/soak_votes_C3.log
ls $O
