#!/bin/bash
# Round 4: the bench lines once more with profiles/round4_pmc.json of the SAME source fingerprint in the tree (label fixes
# in bench.py after the collection: which counter file the instruction counts come from, the rocprofv3 clock only for the
# config it was traced on).  Same binary as collect_round4.sh.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final4b
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu --no-host-leg"
timeout 400 python $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/bench.err
[ -d $R/ab_r3 ] && (cd $R/ab_r3 && timeout 300 python bench.py $Q --steps 20 --warmup 5 2>/dev/null > $O/bench_r3.json)
timeout 300 python $R/bench.py $Q --steps 20 --warmup 5 --no-false-hint-leg 2>/dev/null > $O/bench_arith1.json
timeout 300 python $R/bench.py $Q --steps 20 --warmup 5 --vote-arith 2 --no-false-hint-leg 2>/dev/null > $O/bench_arith2.json
timeout 300 python $R/bench.py $Q --steps 20 --warmup 5 --no-false-hint-leg 2>/dev/null > $O/bench_arith1_b.json
timeout 300 python $R/bench.py $Q --steps 5 --config C1 2>/dev/null > $O/bench_C1.json
timeout 300 python $R/bench.py $Q --steps 5 --config C3 2>/dev/null > $O/bench_C3.json
timeout 300 python $R/bench.py $Q --steps 5 --config C3 --back-tol 2 2>/dev/null > $O/bench_C3_tol2.json
timeout 300 python $R/bench.py $Q --steps 5 --config C4 2>/dev/null > $O/bench_C4.json
ls $O
