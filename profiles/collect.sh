#!/bin/bash
# Collection with the committed binary (round 6 on; parametrised by the round number): the GPU suite, every profiler
# pass the bench line and DESIGN.md quote (tied to the sources by source_fingerprint), the default bench line (all
# configs, clutter, tracked streams), the parity soaks.
#   gpurun --timeout 2400 -- 'bash profiles/collect.sh 6'; then python profiles/install.py 6
# Counter passes: `--kernel-trace` + `--pmc` only, FETCH_SIZE / WRITE_SIZE / SQ in separate runs, mpe:: kernels only.
ROUND=${1:-6}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final$ROUND
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R && timeout 600 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
cd /tmp
python -c "import sys; sys.path.insert(0, '$R'); import rpg_monocular_pose_estimator_amd as m; print(m.source_fingerprint())" > $O/source_fingerprint.txt
Q="--no-cpu --no-host-leg --no-false-hint-leg --headline-only --no-isolated"
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE"
INC='--kernel-include-regex mpe::'
pmc() {  # name, counters, bench args...
  local name=$1 ctr=$2; shift 2
  timeout 240 rocprofv3 --kernel-trace $INC --pmc $ctr --output-format csv -d $O/$name -o p -- python $R/bench.py "$@" > $O/$name.log 2>&1
  timeout 60 python $R/profiles/summarize_pmc_clock.py $O/$name $O/${name}_summary.csv
  find $O/$name -name "*.csv" -delete
}
# the TIMED shape: schedule 6 (72 % of a 32768-frame sub-batch on the rider), side streams taken as concurrent
T6="--steps 3 --warmup 1 $Q --frames 65536 --assume-side-streams"
pmc pmc_fetch FETCH_SIZE $T6
pmc pmc_write WRITE_SIZE $T6
pmc pmc_sq "$SQ" $T6
# one launch shape per kernel (16384 frames, kernels back to back)
T1="--steps 3 --warmup 1 $Q --frames 16384 --pipeline 1"
pmc pmc1_fetch FETCH_SIZE $T1
pmc pmc1_write WRITE_SIZE $T1
pmc pmc1_sq "$SQ" $T1
pmc pmc3_sq "$SQ" $T1 --config C3
pmc pmc3t2_sq "$SQ" $T1 --config C3 --back-tol 2
# the cluttered legs' voting launches (FP64-issue bound) and blob tiers
pmc pmcd4_sq "$SQ" --steps 2 --warmup 1 $Q --frames 32768 --clutter d4 --assume-side-streams
pmc pmcd16_sq "$SQ" --steps 2 --warmup 1 $Q --frames 16384 --clutter d16 --assume-side-streams
pmc pmcsalt_sq "$SQ" --steps 2 --warmup 1 $Q --frames 32768 --clutter salt --assume-side-streams
# the other resolutions' image pass (traffic of the C1 / C4 legs)
pmc pmcC4_fetch FETCH_SIZE --steps 3 --warmup 1 $Q --frames 16384 --config C4 --assume-side-streams
pmc pmcC4_write WRITE_SIZE --steps 3 --warmup 1 $Q --frames 16384 --config C4 --assume-side-streams
pmc pmcC1_fetch FETCH_SIZE --steps 3 --warmup 1 $Q --frames 65536 --config C1 --assume-side-streams
pmc pmcC1_write WRITE_SIZE --steps 3 --warmup 1 $Q --frames 65536 --config C1 --assume-side-streams
# the tracked frame's three kernels: instructions and wave cycles per launch (the dependency-chain count of DESIGN.md 1)
timeout 240 rocprofv3 --kernel-trace $INC --pmc $SQ --output-format csv -d $O/pmc_track -o p -- python $R/bench_streams.py --streams 1 --frames 200 > $O/pmc_track.log 2>&1
timeout 60 python $R/profiles/summarize_pmc_clock.py $O/pmc_track $O/pmc_track_summary.csv
find $O/pmc_track -name "*.csv" -delete
# ---- rocprofv3 --stats: all kernels, and ONLY the dominant kernel traced (the tracer then stretches the schedule less)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/bench.py $Q --steps 20 --warmup 5 > $O/stats.log 2>&1
timeout 60 python $R/profiles/summarize_period.py $O/stats $O/period_summary.json > /dev/null
timeout 300 rocprofv3 --kernel-trace --kernel-include-regex '.*k2_vote<true.*' --stats --output-format csv -d $O/stats_vote -o s -- python $R/bench.py $Q --steps 20 --warmup 5 > $O/stats_vote.log 2>&1
timeout 60 python $R/profiles/summarize_vote_trace.py $O/stats_vote 8 $O/vote_trace_summary.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3 -o s -- python $R/bench.py $Q --steps 5 --config C3 --frames 16384 > $O/stats_c3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_salt -o s -- python $R/bench.py $Q --steps 5 --clutter salt --frames 32768 > $O/stats_salt.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_streams1 -o s -- python $R/bench_streams.py --streams 1 --frames 400 > $O/stats_streams1.log 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
# ---- the counter file of THIS collection, installed on the box (the same command is run again at home on the merged
#      gpurun_out/ and produces the same file): the bench line below then carries traffic / VALU fractions from
#      counters of the binary it times (source_fingerprint match True)
python $R/profiles/install.py $ROUND > $O/install_on_box.log 2>&1
# ---- the default bench line, as the driver runs it
( time timeout 400 python $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/bench.err ) 2>> $O/bench.err
# same box A/B: headline alone; default (vote_arith 3) vs exact powers (1) vs the fast arithmetic alone (2)
QB="--no-cpu --no-host-leg --no-false-hint-leg --headline-only"
timeout 200 python $R/bench.py $QB --steps 20 --warmup 5 2>/dev/null > $O/bench_headline_only.json
timeout 200 python $R/bench.py $QB --steps 20 --warmup 5 --vote-arith 1 2>/dev/null > $O/bench_arith1.json
timeout 200 python $R/bench.py $QB --steps 20 --warmup 5 --vote-arith 2 2>/dev/null > $O/bench_arith2.json
# ---- soaks (every mismatch saved, attributed, classified)
cd $R
timeout 900 python tests/soak_parity.py 524288 C2 65536 gpurun_out/final$ROUND/soak_parity_C2 > $O/soak_parity_C2.log 2>&1; echo "rc $?" >> $O/soak_parity_C2.log
timeout 400 python tests/soak_votes_arith.py 65536 C2 $O/soak_votes_arith_C2.json > $O/soak_votes_arith_C2.log 2>&1; echo "rc $?" >> $O/soak_votes_arith_C2.log
timeout 900 python tests/soak_votes_arith.py 16384 C3 $O/soak_votes_arith_C3.json > $O/soak_votes_arith_C3.log 2>&1; echo "rc $?" >> $O/soak_votes_arith_C3.log
MPE_SOAK_ORACLE=0 MPE_SOAK_PAIR=3,4 timeout 400 python tests/soak_votes.py 1048576 C2 $O/soak_34_C2 > $O/soak_34_C2.log 2>&1; echo "rc $?" >> $O/soak_34_C2.log
MPE_SOAK_ORACLE=0 MPE_SOAK_PAIR=3,4 MPE_SOAK_STRICT_FRAMES=65536 timeout 400 python tests/soak_votes.py 65536 C3 $O/soak_34_C3 > $O/soak_34_C3.log 2>&1; echo "rc $?" >> $O/soak_34_C3.log
MPE_SOAK_ORACLE=0 MPE_SOAK_PAIR=3,4 MPE_SOAK_DISTRACTORS=16 timeout 400 python tests/soak_votes.py 32768 C2 $O/soak_34_d16 > $O/soak_34_d16.log 2>&1; echo "rc $?" >> $O/soak_34_d16.log
MPE_SOAK_ORACLE=0 MPE_SOAK_PAIR=3,4 MPE_SOAK_DISTRACTORS=4 timeout 400 python tests/soak_votes.py 262144 C2 $O/soak_34_d4 > $O/soak_34_d4.log 2>&1; echo "rc $?" >> $O/soak_34_d4.log
timeout 400 python tests/soak_parity.py 4096 C3 2048 gpurun_out/final$ROUND/soak_parity_C3 > $O/soak_parity_C3.log 2>&1; echo "rc $?" >> $O/soak_parity_C3.log
timeout 400 python tests/soak_parity.py 8192 C4 4096 gpurun_out/final$ROUND/soak_parity_C4 > $O/soak_parity_C4.log 2>&1; echo "rc $?" >> $O/soak_parity_C4.log
timeout 400 python tests/soak_parity.py 32768 C1 32768 gpurun_out/final$ROUND/soak_parity_C1 > $O/soak_parity_C1.log 2>&1; echo "rc $?" >> $O/soak_parity_C1.log
MPE_SOAK_CLUTTER=salt timeout 400 python tests/soak_parity.py 32768 C2 8192 gpurun_out/final$ROUND/soak_parity_clutter_salt > $O/soak_parity_clutter_salt.log 2>&1; echo "rc $?" >> $O/soak_parity_clutter_salt.log
MPE_SOAK_CLUTTER=d4 timeout 400 python tests/soak_parity.py 65536 C2 16384 gpurun_out/final$ROUND/soak_parity_clutter_d4 > $O/soak_parity_clutter_d4.log 2>&1; echo "rc $?" >> $O/soak_parity_clutter_d4.log
MPE_SOAK_CLUTTER=d16 timeout 600 python tests/soak_parity.py 8192 C2 4096 gpurun_out/final$ROUND/soak_parity_clutter_d16 > $O/soak_parity_clutter_d16.log 2>&1; echo "rc $?" >> $O/soak_parity_clutter_d16.log
MPE_SOAK_CLUTTER=salt timeout 600 python tests/soak_parity.py 4096 C4 2048 gpurun_out/final$ROUND/soak_parity_clutter_salt_C4 > $O/soak_parity_clutter_salt_C4.log 2>&1; echo "rc $?" >> $O/soak_parity_clutter_salt_C4.log
MPE_SOAK_CLUTTER=d4 timeout 600 python tests/soak_parity.py 4096 C3 2048 gpurun_out/final$ROUND/soak_parity_clutter_d4_C3 > $O/soak_parity_clutter_d4_C3.log 2>&1; echo "rc $?" >> $O/soak_parity_clutter_d4_C3.log
timeout 400 python tests/soak_general_tier.py 240 gpurun_out/final$ROUND/soak_general_tier > $O/soak_general_tier.log 2>&1; echo "rc $?" >> $O/soak_general_tier.log
timeout 400 python tests/soak_tracking.py 128 160 C2 gpurun_out/final$ROUND/soak_tracking > $O/soak_tracking.log 2>&1; echo "rc $?" >> $O/soak_tracking.log
MPE_SOAK_SALT=0.003 timeout 600 python tests/soak_tracking.py 128 160 C2 gpurun_out/final$ROUND/soak_tracking_salt > $O/soak_tracking_salt.log 2>&1; echo "rc $?" >> $O/soak_tracking_salt.log
ls $O
