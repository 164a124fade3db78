#!/bin/bash
# round 3: schedule 7 (votes back to back, blobs beside them) against schedule 6 on one box
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3k
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R && timeout 400 python -m pytest tests/test_gpu_parity_large.py -m gpu -x -q -k "deep_schedule or streaming" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
cd /tmp
for m in 6 7 6 7; do
timeout 150 python $R/bench.py --no-cpu --no-host-leg --steps 20 --warmup 5 --pipeline-mode $m > $O/bench_m${m}_$RANDOM.json 2>>$O/bench.err
done
timeout 150 python $R/bench.py --no-cpu --no-host-leg --steps 20 --warmup 5 --pipeline-mode 7 --pipeline 16 --frames 262144 > $O/bench_m7_p16.json 2>>$O/bench.err
ls $O
