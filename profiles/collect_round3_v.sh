#!/bin/bash
# round 3: narrow grid for the (normally empty) follow-up blob tier at <= 5 markers, against the stock build, same box
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3v
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in new stock new stock new stock; do
  L=$R/rpg_monocular_pose_estimator_amd/libmpe_hip.so
  [ $v = stock ] && L=$R/rpg_monocular_pose_estimator_amd/variants/libmpe_stock.so
  MPE_LIB=$L timeout 150 python $R/bench.py --no-cpu --no-host-leg --steps 20 --warmup 5 > $O/bench_${v}_$RANDOM.json 2>>$O/bench.err
done
ls $O
