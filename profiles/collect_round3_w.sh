#!/bin/bash
# round 3: C3 (8 LEDs, 12 detections) end-to-end soak at back-projection tolerance 2 px, where most frames initialise:
# 16 384 frames of the final binary against the oracle, with forensics
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final3
mkdir -p $O
cd $R
MPE_VOTE_ARITH=1 MPE_BACK_TOL=2 timeout 900 python tests/soak_parity.py 16384 C3 4096 gpurun_out/final3/soak_parity_C3_tol2_16k 2>/dev/null | tail -1 > $O/soak_fast_c3_tol2_16k.json; echo "rc $?" > $O/soak_c3_tol2_16k.rc
cat $O/soak_c3_tol2_16k.rc
