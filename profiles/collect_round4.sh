#!/bin/bash
# Round 4, FINAL collection with the committed binary: the GPU suite, the bench line, every profiler pass the bench line
# and DESIGN.md quote (tied to the sources by source_fingerprint), the other configs, tracked-frame benches, soaks.
#   gpurun --timeout 2400 -- 'bash profiles/collect_round4.sh'; then python profiles/install_round4.py
# Counter passes: `--kernel-trace` + `--pmc` only, FETCH_SIZE / WRITE_SIZE / SQ in separate runs, mpe:: kernels only.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final4
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R && timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
cd /tmp
python -c "import sys; sys.path.insert(0, '$R'); import rpg_monocular_pose_estimator_amd as m; print(m.source_fingerprint())" > $O/source_fingerprint.txt
Q="--no-cpu --no-host-leg"
# ---- counter passes first
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE"
INC='--kernel-include-regex mpe::'
pmc() {  # name, counters, bench args...
  local name=$1 ctr=$2; shift 2
  timeout 300 rocprofv3 --kernel-trace $INC --pmc $ctr --output-format csv -d $O/$name -o p -- python $R/bench.py "$@" > $O/$name.log 2>&1
  timeout 60 python $R/profiles/summarize_pmc_clock.py $O/$name $O/${name}_summary.csv
  find $O/$name -name "*.csv" -delete
}
# the TIMED shape: schedule 6 (70 % of a 32768-frame sub-batch on the rider), side streams taken as concurrent
T6="--steps 3 --warmup 1 $Q --no-false-hint-leg --frames 65536 --assume-side-streams"
pmc pmc_fetch FETCH_SIZE $T6
pmc pmc_write WRITE_SIZE $T6
pmc pmc_sq "$SQ" $T6
# one launch shape per kernel (16384 frames, kernels back to back)
T1="--steps 3 --warmup 1 $Q --no-false-hint-leg --frames 16384 --pipeline 1"
pmc pmc1_fetch FETCH_SIZE $T1
pmc pmc1_write WRITE_SIZE $T1
pmc pmc1_sq "$SQ" $T1
pmc pmc3_sq "$SQ" $T1 --config C3
# the other resolutions' image pass (traffic of the C1 / C4 bench records)
pmc pmcC4_fetch FETCH_SIZE --steps 3 --warmup 1 $Q --no-false-hint-leg --frames 16384 --config C4 --assume-side-streams
pmc pmcC4_write WRITE_SIZE --steps 3 --warmup 1 $Q --no-false-hint-leg --frames 16384 --config C4 --assume-side-streams
pmc pmcC1_fetch FETCH_SIZE --steps 3 --warmup 1 $Q --no-false-hint-leg --frames 65536 --config C1 --assume-side-streams
pmc pmcC1_write WRITE_SIZE --steps 3 --warmup 1 $Q --no-false-hint-leg --frames 65536 --config C1 --assume-side-streams
# ---- rocprofv3 --stats: all kernels, and ONLY the dominant kernel traced (the tracer then stretches the schedule less)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/bench.py $Q --steps 20 --warmup 5 --no-false-hint-leg > $O/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --kernel-include-regex 'k2_vote<true' --stats --output-format csv -d $O/stats_vote -o s -- python $R/bench.py $Q --steps 20 --warmup 5 --no-false-hint-leg > $O/stats_vote.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_seq -o s -- python $R/bench.py $Q --no-false-hint-leg --pipeline 1 --frames 16384 --steps 20 --warmup 3 > $O/stats_seq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3 -o s -- python $R/bench.py $Q --steps 5 --config C3 --frames 65536 > $O/stats_c3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_streams1 -o s -- python $R/bench_streams.py --streams 1 --frames 400 > $O/stats_streams1.log 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
# ---- the counter file of THIS collection, installed on the box (the same command is run again at home on the merged
#      gpurun_out/ and produces the same file): the bench line below then carries roofline.traffic / frac_rocprofv3 from
#      counters of the binary it times (source_fingerprint match True)
python $R/profiles/install_round4.py > $O/install_on_box.log 2>&1
# ---- bench lines
timeout 400 python $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/bench.err
# the round-3 tree on the same box (ab_r3/: commit 14e6180 built in place; not part of the repository, travels with gpurun)
[ -d $R/ab_r3 ] && (cd $R/ab_r3 && timeout 300 python bench.py $Q --steps 20 --warmup 5 2>/dev/null > $O/bench_r3.json)
timeout 300 python $R/bench.py $Q --steps 20 --warmup 5 --vote-arith 2 --no-false-hint-leg 2>/dev/null > $O/bench_arith2.json
timeout 300 python $R/bench.py $Q --steps 20 --warmup 5 --no-false-hint-leg 2>/dev/null > $O/bench_arith1.json
timeout 300 python $R/bench.py $Q --steps 5 --config C1 --frames 65536 2>/dev/null > $O/bench_C1.json
timeout 300 python $R/bench.py $Q --steps 5 --config C3 --frames 65536 2>/dev/null > $O/bench_C3.json
timeout 300 python $R/bench.py $Q --steps 5 --config C3 --frames 65536 --back-tol 2 2>/dev/null > $O/bench_C3_tol2.json
timeout 300 python $R/bench.py $Q --steps 5 --config C4 --frames 16384 2>/dev/null > $O/bench_C4.json
timeout 300 python $R/bench_streams.py --streams 1 --frames 400 2>/dev/null | tail -1 > $O/streams1.json
timeout 300 python $R/bench_streams.py --streams 8 --frames 400 2>/dev/null | tail -1 > $O/streams8.json
for n in 64; do timeout 200 python $R/bench_streams.py --streams $n --frames 300 --lockstep 2>/dev/null | tail -1 > $O/lockstep$n.json; done
timeout 300 python $R/bench_streams.py --streams 512 --frames 300 --lockstep --groups 8 --group-threads 8 2>/dev/null | tail -1 > $O/lockstep512g8t8.json
# ---- soaks (every mismatch saved, attributed, classified; default vs strict histograms must be identical)
cd $R
timeout 900 python tests/soak_votes.py 131072 C2 gpurun_out/final4/soak_votes_C2 > $O/soak_votes_C2.log 2>&1; echo "rc $?" >> $O/soak_votes_C2.log
timeout 900 python tests/soak_votes.py 4096 C3 gpurun_out/final4/soak_votes_C3 > $O/soak_votes_C3.log 2>&1; echo "rc $?" >> $O/soak_votes_C3.log
timeout 900 python tests/soak_parity.py 131072 C2 65536 gpurun_out/final4/soak_parity_C2 > $O/soak_parity_C2.log 2>&1; echo "rc $?" >> $O/soak_parity_C2.log
timeout 600 python tests/soak_parity.py 4096 C3 2048 gpurun_out/final4/soak_parity_C3 > $O/soak_parity_C3.log 2>&1; echo "rc $?" >> $O/soak_parity_C3.log
MPE_BACK_TOL=2 timeout 600 python tests/soak_parity.py 4096 C3 2048 gpurun_out/final4/soak_parity_C3_tol2 > $O/soak_parity_C3_tol2.log 2>&1; echo "rc $?" >> $O/soak_parity_C3_tol2.log
timeout 600 python tests/soak_parity.py 8192 C4 4096 gpurun_out/final4/soak_parity_C4 > $O/soak_parity_C4.log 2>&1; echo "rc $?" >> $O/soak_parity_C4.log
timeout 600 python tests/soak_parity.py 32768 C1 32768 gpurun_out/final4/soak_parity_C1 > $O/soak_parity_C1.log 2>&1; echo "rc $?" >> $O/soak_parity_C1.log
timeout 600 python tests/soak_tracking.py 256 160 C2 gpurun_out/final4/soak_tracking > $O/soak_tracking.log 2>&1; echo "rc $?" >> $O/soak_tracking.log
ls $O
