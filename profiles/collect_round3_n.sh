#!/bin/bash
# round 3: early blobs (schedule 6) A/B on one box + the GPU tests that cover the schedules
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3n
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R && timeout 900 python -m pytest tests -m gpu -x -q -k "full_size or streaming or overflow or deep or pipelin or schedule or capacity" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
cd /tmp
for e in 0 1 0 1; do
timeout 150 python $R/bench.py --no-cpu --no-host-leg --steps 20 --warmup 5 --early-blobs $e > $O/bench_e${e}_$RANDOM.json 2>>$O/bench.err
done
for sp in 40 50; do
timeout 150 python $R/bench.py --no-cpu --no-host-leg --steps 20 --warmup 5 --early-blobs 1 --scan-split-pct $sp > $O/bench_e1_split$sp.json 2>>$O/bench.err
done
timeout 150 python $R/bench.py --no-cpu --no-host-leg --steps 20 --warmup 5 --early-blobs 1 --scan-split-pct 40 --side-scan-blocks 4 > $O/bench_e1_split40_b4.json 2>>$O/bench.err
ls $O
