#!/bin/bash
# round 3: one more end-to-end C2 soak of the final binary with forensics — 1 048 576 frames, fast arithmetic
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final3
mkdir -p $O
cd $R
MPE_VOTE_ARITH=1 timeout 1500 python tests/soak_parity.py 1048576 C2 65536 gpurun_out/final3/soak_parity_C2_fast_1M 2>/dev/null | tail -1 > $O/soak_fast_1m.json; echo "rc $?" > $O/soak_fast_1m.rc
cat $O/soak_fast_1m.rc
