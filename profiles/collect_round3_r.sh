#!/bin/bash
# round 3: tail stream at the lowest stream priority, with the new and the previous k3a_validate, same box
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3r
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
V=$R/rpg_monocular_pose_estimator_amd/variants
for v in new newprio prev prevprio new newprio prev prevprio; do
  L=$R/rpg_monocular_pose_estimator_amd/libmpe_hip.so
  [ $v != new ] && L=$V/libmpe_$v.so
  MPE_LIB=$L timeout 150 python $R/bench.py --no-cpu --no-host-leg --steps 20 --warmup 5 > $O/bench_${v}_$RANDOM.json 2>>$O/bench.err
done
ls $O
