#!/bin/bash
# Round 4, first GPU call: the GPU suite with the new default voting arithmetic (fast kernel + k2_vote_fixup), the bench
# line, a same-box A/B against the fast arithmetic alone (vote_arith 2 = round 3's default) and the vote-histogram
# soaks (default vs strict vs fast alone vs oracle) at C2 and C3.
#   gpurun --timeout 2700 -- 'bash profiles/collect_round4_a.sh'
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4a
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R && timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
cd /tmp
timeout 400 python $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/bench.err
for i in 1 2; do
timeout 300 python $R/bench.py --no-cpu --no-host-leg --steps 20 --warmup 5 --vote-arith 2 --no-false-hint-leg 2>/dev/null > $O/bench_arith2_$i.json
timeout 300 python $R/bench.py --no-cpu --no-host-leg --steps 20 --warmup 5 --no-false-hint-leg 2>/dev/null > $O/bench_arith1_$i.json
done
timeout 300 python $R/bench.py --no-cpu --no-host-leg --steps 5 --config C3 --frames 65536 2>$O/bench_C3.err > $O/bench_C3.json
timeout 300 python $R/bench.py --no-cpu --no-host-leg --steps 5 --config C3 --frames 65536 --vote-arith 2 2>/dev/null > $O/bench_C3_arith2.json
cd $R
timeout 900 python tests/soak_votes.py 131072 C2 gpurun_out/r4a/soak_votes_C2 > $O/soak_votes_C2.log 2>&1; echo "rc $?" >> $O/soak_votes_C2.log
timeout 900 python tests/soak_votes.py 4096 C3 gpurun_out/r4a/soak_votes_C3 > $O/soak_votes_C3.log 2>&1; echo "rc $?" >> $O/soak_votes_C3.log
ls $O
