#!/bin/bash
# Round 4, fifth GPU call: scan-split sweep with the faster blob kernel, tracked-frame benches, vote soaks with the
# narrower band.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4e
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--no-cpu --no-host-leg --steps 20 --warmup 5 --no-false-hint-leg"
timeout 300 python $R/bench.py $B 2>/dev/null > $O/bench_30_3.json
for pct in 20 25 35 40; do timeout 300 python $R/bench.py $B --scan-split-pct $pct 2>/dev/null > $O/bench_${pct}_3.json; done
for blk in 2 4; do timeout 300 python $R/bench.py $B --scan-split-pct 35 --side-scan-blocks $blk 2>/dev/null > $O/bench_35_$blk.json; done
timeout 300 python $R/bench.py $B 2>/dev/null > $O/bench_30_3_again.json
(cd $R/ab_r3 && timeout 300 python bench.py --no-cpu --no-host-leg --steps 20 --warmup 5 2>/dev/null > $O/bench_r3.json)
timeout 300 python $R/bench_streams.py --streams 1 --frames 400 2>/dev/null | tail -1 > $O/streams1.json
timeout 300 python $R/bench_streams.py --streams 8 --frames 400 2>/dev/null | tail -1 > $O/streams8.json
for n in 8 64; do timeout 200 python $R/bench_streams.py --streams $n --frames 300 --lockstep 2>/dev/null | tail -1 > $O/lockstep$n.json; done
(cd $R/ab_r3 && timeout 300 python bench_streams.py --streams 1 --frames 400 2>/dev/null | tail -1 > $O/streams1_r3.json)
timeout 300 python $R/bench.py --no-cpu --no-host-leg --steps 5 --config C3 --frames 65536 2>/dev/null > $O/bench_C3.json
cd $R
timeout 900 python tests/soak_votes.py 32768 C2 gpurun_out/r4e/soak_votes_C2 > $O/soak_votes_C2.log 2>&1; echo "rc $?" >> $O/soak_votes_C2.log
timeout 900 python tests/soak_votes.py 2048 C3 gpurun_out/r4e/soak_votes_C3 > $O/soak_votes_C3.log 2>&1; echo "rc $?" >> $O/soak_votes_C3.log
ls $O
