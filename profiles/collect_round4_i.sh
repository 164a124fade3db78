#!/bin/bash
# Round 4, ninth GPU call (queue flushed in full passes after every root, disc-shaped grid dilation): the plain voting kernel's per-root part in single precision (M = G K T^T Rm, grid cells by v_cvt_pk_u8_f32, exact evaluation
# one entry per lane) at C3 — GPU suite, C3 A/B against the round-3 tree, soaks.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4i
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R && timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
cd /tmp
timeout 300 python $R/bench.py --no-cpu --no-host-leg --steps 5 --config C3 --frames 65536 2>$O/bench_C3.err > $O/bench_C3.json
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE"
timeout 300 rocprofv3 --kernel-trace --kernel-include-regex mpe:: --pmc $SQ --output-format csv -d $O/pmc3_sq -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-host-leg --frames 16384 --pipeline 1 --config C3 > $O/pmc3_sq.log 2>&1
timeout 60 python $R/profiles/summarize_pmc_clock.py $O/pmc3_sq $O/pmc3_sq_summary.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
cd $R
timeout 900 python tests/soak_votes.py 4096 C3 gpurun_out/r4i/soak_votes_C3 > $O/soak_votes_C3.log 2>&1; echo "rc $?" >> $O/soak_votes_C3.log
MPE_BACK_TOL=5 timeout 600 python tests/soak_parity.py 4096 C3 2048 gpurun_out/r4i/soak_parity_C3 > $O/soak_parity_C3.log 2>&1; echo "rc $?" >> $O/soak_parity_C3.log
MPE_BACK_TOL=2 timeout 300 python $R/bench.py --no-cpu --no-host-leg --steps 5 --config C3 --frames 65536 --back-tol 2 2>/dev/null > $O/bench_C3_tol2.json
timeout 300 python $R/bench.py --no-cpu --no-host-leg --steps 20 --warmup 5 --no-false-hint-leg 2>/dev/null > $O/bench_C2.json
timeout 300 python $R/bench.py --no-cpu --no-host-leg --steps 10 --config C4 2>/dev/null > $O/bench_C4.json
ls $O
