#!/bin/bash
# round 3: k3a_validate with a smaller register budget (3 / 4 waves per SIMD, spilling) beside the stock build — does the
# blob window get shorter when the tail's 241-VGPR waves stop halving a SIMD's blob occupancy?
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3o
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in stock k3a3 k3a4 stock k3a3 k3a4; do
  L=$R/rpg_monocular_pose_estimator_amd/libmpe_hip.so
  [ $v != stock ] && L=$R/rpg_monocular_pose_estimator_amd/variants/libmpe_$v.so
  MPE_LIB=$L timeout 150 python $R/bench.py --no-cpu --no-host-leg --steps 20 --warmup 5 > $O/bench_${v}_$RANDOM.json 2>>$O/bench.err
done
ls $O
