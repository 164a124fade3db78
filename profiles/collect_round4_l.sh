#!/bin/bash
# Round 4, twelfth GPU call (scan-carrying voting kernel single precision behind the cancellations; island windows of one word): the blob kernel's phases, timed by experiment builds that end a frame's work after phase n
# (profiles/build_k1b_stops.sh -> build_variants/), kernels back to back (--pipeline 1), 65 536 C2 frames.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4l
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for n in 4 5 full; do
  LIB=$R/build_variants/libmpe_hip_stop$n.so
  [ $n = full ] && LIB=$R/rpg_monocular_pose_estimator_amd/libmpe_hip.so
  MPE_LIB=$LIB timeout 200 python $R/bench.py --no-cpu --no-host-leg --pipeline 1 --frames 65536 --steps 10 --warmup 3 2>$O/err_$n.log | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$n', json.dumps(d.get('kernel_ms')), d.get('ms_per_step'))" >> $O/phases.log 2>&1
done
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE"
for n in 4 full; do
  LIB=$R/build_variants/libmpe_hip_stop$n.so
  [ $n = full ] && LIB=$R/rpg_monocular_pose_estimator_amd/libmpe_hip.so
  MPE_LIB=$LIB timeout 200 rocprofv3 --kernel-trace --kernel-include-regex 'k1b_blobs' --pmc $SQ --output-format csv -d $O/pmc_$n -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-host-leg --frames 16384 --pipeline 1 > $O/pmc_$n.log 2>&1
  timeout 60 python $R/profiles/summarize_pmc_clock.py $O/pmc_$n $O/pmc_${n}_summary.csv
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
cd $R && timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
cd /tmp
timeout 300 python $R/bench.py --no-cpu --no-host-leg --steps 20 --warmup 5 --no-false-hint-leg 2>/dev/null > $O/bench_C2.json
(cd $R/ab_r3 && timeout 300 python bench.py --no-cpu --no-host-leg --steps 20 --warmup 5 2>/dev/null > $O/bench_C2_r3.json)
timeout 300 python $R/bench.py --no-cpu --no-host-leg --steps 20 --warmup 5 --no-false-hint-leg 2>/dev/null > $O/bench_C2_b.json
timeout 200 python $R/bench_streams.py --streams 1 --frames 400 2>/dev/null > $O/streams1.json
timeout 200 python $R/bench_streams.py --streams 64 --frames 300 --lockstep 2>/dev/null > $O/lockstep64.json
cd $R
timeout 900 python tests/soak_votes.py 65536 C2 gpurun_out/r4l/soak_votes_C2 > $O/soak_votes_C2.log 2>&1; echo "rc $?" >> $O/soak_votes_C2.log
timeout 900 python tests/soak_votes.py 2048 C3 gpurun_out/r4l/soak_votes_C3 > $O/soak_votes_C3.log 2>&1; echo "rc $?" >> $O/soak_votes_C3.log
timeout 600 python tests/soak_parity.py 131072 C2 65536 gpurun_out/r4l/soak_parity_C2 > $O/soak_parity_C2.log 2>&1; echo "rc $?" >> $O/soak_parity_C2.log
cat $O/phases.log
