#!/bin/bash
# round 3: C3 A/B on one box — table slices in LDS with full-size blocks against the un-sliced kernel
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3j
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
timeout 200 python $R/bench.py --no-cpu --no-host-leg --steps 5 --config C3 --frames 65536 --vote-splits $v 2>/dev/null > $O/bench_C3_splits$v.json
done
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE"
for v in 0 1; do
timeout 200 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/pmc3_sq$v -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu --no-host-leg --frames 16384 --pipeline 1 --config C3 --vote-splits $v > $O/pmc3_sq$v.log 2>&1
timeout 60 python $R/profiles/summarize_pmc_clock.py $O/pmc3_sq$v $O/pmc3_sq${v}_summary.csv
find $O/pmc3_sq$v -name "*.csv" -delete
done
cd $R && timeout 300 python -m pytest tests/test_gpu_parity_large.py -m gpu -x -q -k "C3 or c3 or many_markers" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
ls $O
