#!/bin/bash
# (first half; the counter passes of the fused launch shape crashed in this script — see collect_round3_final_b.sh)
# Collects everything under profiles/round3_* in ONE gpurun call on a 1xMI355X box:
#   gpurun --timeout 2400 -- 'bash profiles/collect_round3.sh'
# Raw output goes to gpurun_out/final3/; profiles/install_round3.py condenses it into the committed files.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final3
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0, '$R'); import rpg_monocular_pose_estimator_amd as m; print(m.source_fingerprint())" > $O/source_fingerprint.txt
timeout 300 python $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/bench.err
timeout 300 python $R/bench.py --no-cpu --no-host-leg --steps 20 --warmup 5 --no-streaming 2>/dev/null > $O/bench_nostream.json
for c in C1 C3 C4; do timeout 300 python $R/bench.py --no-cpu --no-host-leg --steps 5 --config $c --frames $([ $c = C4 ] && echo 16384 || echo 65536) 2>/dev/null > $O/bench_$c.json; done
timeout 300 python $R/bench_streams.py --streams 1 --frames 400 2>/dev/null | tail -1 > $O/streams1.json
timeout 300 python $R/bench_streams.py --streams 8 --frames 400 2>/dev/null | tail -1 > $O/streams8.json
for n in 8 64 256; do timeout 200 python $R/bench_streams.py --streams $n --frames 300 --lockstep 2>/dev/null | tail -1 > $O/lockstep$n.json; done
timeout 300 python $R/bench_streams.py --streams 256 --frames 300 --lockstep --groups 4 --group-threads 4 2>/dev/null | tail -1 > $O/lockstep256g4t4.json
timeout 300 python $R/bench_streams.py --streams 512 --frames 300 --lockstep --groups 8 --group-threads 8 2>/dev/null | tail -1 > $O/lockstep512g8t8.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/bench.py --no-cpu --no-host-leg --steps 20 --warmup 5 > $O/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_seq -o s -- python $R/bench.py --no-cpu --no-host-leg --pipeline 1 --frames 16384 --steps 20 --warmup 3 > $O/stats_seq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3 -o s -- python $R/bench.py --no-cpu --no-host-leg --config C3 --pipeline 1 --frames 16384 --steps 3 --warmup 1 > $O/stats_c3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_lockstep -o s -- python $R/bench_streams.py --streams 64 --frames 300 --lockstep > $O/stats_lockstep.log 2>&1
find $O -name "*kernel_trace.csv" -delete
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE"
# the fused launch shape of the timed run: 32768 frames per launch, every voting launch carries a scan (streaming)
ARGS="--steps 3 --warmup 1 --no-cpu --no-host-leg --frames 65536"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python $R/bench.py $ARGS > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python $R/bench.py $ARGS > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/pmc_sq -o p -- python $R/bench.py $ARGS > $O/pmc_sq.log 2>&1
# the stand-alone kernels at ONE launch shape (16384 frames, one launch per step)
ARGS1="--steps 3 --warmup 1 --no-cpu --no-host-leg --frames 16384 --pipeline 1"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc1_fetch -o p -- python $R/bench.py $ARGS1 > $O/pmc1_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc1_write -o p -- python $R/bench.py $ARGS1 > $O/pmc1_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/pmc1_sq -o p -- python $R/bench.py $ARGS1 > $O/pmc1_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/pmc3_sq -o p -- python $R/bench.py $ARGS1 --config C3 > $O/pmc3_sq.log 2>&1
for n in pmc_fetch pmc_write pmc_sq pmc1_fetch pmc1_write pmc1_sq pmc3_sq; do
  python $R/profiles/summarize_pmc_clock.py $O/$n $O/${n}_summary.csv
done
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
# parity soaks with forensics (every mismatch saved + classified; exit code 1 = an unexplained one)
cd $R
timeout 600 python tests/soak_votes.py 131072 C2 gpurun_out/final3/soak_votes_C2 2>/dev/null | tail -1 > $O/soak_votes.json; echo "rc $?" > $O/soak_votes.rc
MPE_VOTE_ARITH=1 timeout 900 python tests/soak_parity.py 524288 C2 65536 gpurun_out/final3/soak_parity_C2_fast 2>/dev/null | tail -1 > $O/soak_fast.json; echo "rc $?" > $O/soak_fast.rc
MPE_VOTE_ARITH=0 timeout 900 python tests/soak_parity.py 131072 C2 65536 gpurun_out/final3/soak_parity_C2_strict 2>/dev/null | tail -1 > $O/soak_strict.json; echo "rc $?" > $O/soak_strict.rc
MPE_VOTE_ARITH=1 timeout 900 python tests/soak_parity.py 16384 C3 4096 gpurun_out/final3/soak_parity_C3 2>/dev/null | tail -1 > $O/soak_fast_c3.json; echo "rc $?" > $O/soak_c3.rc
MPE_VOTE_ARITH=1 MPE_BACK_TOL=2 timeout 600 python tests/soak_parity.py 4096 C3 4096 gpurun_out/final3/soak_parity_C3_tol2 2>/dev/null | tail -1 > $O/soak_fast_c3_tol2.json
MPE_VOTE_ARITH=1 timeout 900 python tests/soak_parity.py 16384 C4 4096 gpurun_out/final3/soak_parity_C4 2>/dev/null | tail -1 > $O/soak_fast_c4.json
MPE_VOTE_ARITH=1 timeout 900 python tests/soak_parity.py 65536 C1 65536 gpurun_out/final3/soak_parity_C1 2>/dev/null | tail -1 > $O/soak_fast_c1.json
ls $O
