#!/bin/bash
# round 3: k3a_validate with the histogram's column maxima found by the group's lanes (instead of lane 0 scanning the
# histogram n_m times from global memory) against the previous build, same box; GPU suite first
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3q
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R && timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
cd /tmp
for v in new prev new prev; do
  L=$R/rpg_monocular_pose_estimator_amd/libmpe_hip.so
  [ $v = prev ] && L=$R/rpg_monocular_pose_estimator_amd/variants/libmpe_prev.so
  MPE_LIB=$L timeout 150 python $R/bench.py --no-cpu --no-host-leg --steps 20 --warmup 5 > $O/bench_${v}_$RANDOM.json 2>>$O/bench.err
done
MPE_LIB=$R/rpg_monocular_pose_estimator_amd/variants/libmpe_prev.so timeout 150 python $R/bench.py --no-cpu --no-host-leg --steps 5 --no-streaming > $O/bench_prev_nostream.json 2>>$O/bench.err
timeout 150 python $R/bench.py --no-cpu --no-host-leg --steps 5 --no-streaming > $O/bench_new_nostream.json 2>>$O/bench.err
ls $O
