#!/bin/bash
# round 3: tracking-path parity soak — 256 streams x 160 frames in lock step against the oracle's state machine
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final3
mkdir -p $O
cd $R
timeout 900 python tests/soak_tracking.py 256 160 C2 gpurun_out/final3/soak_tracking_C2 2>$O/soak_tracking.err | tail -1 > $O/soak_tracking.json; echo "rc $?" > $O/soak_tracking.rc
tail -3 $O/soak_tracking.err
