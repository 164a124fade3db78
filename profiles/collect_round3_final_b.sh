#!/bin/bash
# Second half of the round-3 collection, taken with the FINAL binary (the tracked-frame changes came after
# collect_round3.sh): the bench line, every counter pass the bench line quotes (tied to the sources by
# source_fingerprint), the tracked-frame benches and the whole GPU test-suite.  Counter passes only look at mpe::
# kernels (--kernel-include-regex): collecting counters on torch's thousands of small fill / index kernels of a
# 65 536-frame synthetic batch crashed the profiler in collect_round3.sh.
#   gpurun --timeout 1800 -- 'bash profiles/collect_round3_final_b.sh'; then python profiles/install_round3.py
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R && timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
cd /tmp
python -c "import sys; sys.path.insert(0, '$R'); import rpg_monocular_pose_estimator_amd as m; print(m.source_fingerprint())" > $O/source_fingerprint.txt
timeout 300 python $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/bench.err
timeout 300 python $R/bench.py --no-cpu --no-host-leg --steps 20 --warmup 5 --no-streaming 2>/dev/null > $O/bench_nostream.json
timeout 300 python $R/bench.py --no-cpu --no-host-leg --steps 20 --warmup 5 --no-vote-events 2>/dev/null > $O/bench_no_vote_events.json
timeout 300 python $R/bench.py --no-cpu --no-host-leg --steps 20 --warmup 5 2>/dev/null > $O/bench_vote_events.json
timeout 300 python $R/bench.py --no-cpu --no-host-leg --steps 5 --config C3 --frames 65536 --vote-splits 1 2>/dev/null > $O/bench_C3_unsliced.json
timeout 300 python $R/bench_streams.py --streams 1 --frames 400 2>/dev/null | tail -1 > $O/streams1.json
timeout 300 python $R/bench_streams.py --streams 8 --frames 400 2>/dev/null | tail -1 > $O/streams8.json
for n in 8 64 256; do timeout 200 python $R/bench_streams.py --streams $n --frames 300 --lockstep 2>/dev/null | tail -1 > $O/lockstep$n.json; done
timeout 300 python $R/bench_streams.py --streams 256 --frames 300 --lockstep --groups 4 --group-threads 4 2>/dev/null | tail -1 > $O/lockstep256g4t4.json
timeout 300 python $R/bench_streams.py --streams 512 --frames 300 --lockstep --groups 8 --group-threads 8 2>/dev/null | tail -1 > $O/lockstep512g8t8.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/bench.py --no-cpu --no-host-leg --steps 20 --warmup 5 > $O/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_lockstep -o s -- python $R/bench_streams.py --streams 64 --frames 300 --lockstep > $O/stats_lockstep.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_streams1 -o s -- python $R/bench_streams.py --streams 1 --frames 400 > $O/stats_streams1.log 2>&1
find $O -name "*kernel_trace.csv" -delete
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE"
INC='--kernel-include-regex mpe::'
# the fused launch shape of the timed run: 32768 frames per launch, every voting launch carries a scan (streaming)
ARGS="--steps 3 --warmup 1 --no-cpu --no-host-leg --frames 65536"
timeout 300 rocprofv3 --kernel-trace $INC --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python $R/bench.py $ARGS > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace $INC --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python $R/bench.py $ARGS > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace $INC --pmc $SQ --output-format csv -d $O/pmc_sq -o p -- python $R/bench.py $ARGS > $O/pmc_sq.log 2>&1
# the stand-alone kernels at ONE launch shape (16384 frames, one launch per step)
ARGS1="--steps 3 --warmup 1 --no-cpu --no-host-leg --frames 16384 --pipeline 1"
timeout 300 rocprofv3 --kernel-trace $INC --pmc FETCH_SIZE --output-format csv -d $O/pmc1_fetch -o p -- python $R/bench.py $ARGS1 > $O/pmc1_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace $INC --pmc WRITE_SIZE --output-format csv -d $O/pmc1_write -o p -- python $R/bench.py $ARGS1 > $O/pmc1_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace $INC --pmc $SQ --output-format csv -d $O/pmc1_sq -o p -- python $R/bench.py $ARGS1 > $O/pmc1_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace $INC --pmc $SQ --output-format csv -d $O/pmc3_sq -o p -- python $R/bench.py $ARGS1 --config C3 > $O/pmc3_sq.log 2>&1
for n in pmc_fetch pmc_write pmc_sq pmc1_fetch pmc1_write pmc1_sq pmc3_sq; do
  timeout 60 python $R/profiles/summarize_pmc_clock.py $O/$n $O/${n}_summary.csv
done
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
ls $O
