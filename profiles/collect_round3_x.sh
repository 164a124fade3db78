#!/bin/bash
# round 3: vote histograms at C3 (8 markers, 12 detections, 73 920 hypotheses per frame) against the oracle, both arithmetics
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final3
mkdir -p $O
cd $R
timeout 600 python tests/soak_votes.py 4096 C3 gpurun_out/final3/soak_votes_C3 2>/dev/null | tail -1 > $O/soak_votes_c3.json; echo "rc $?" > $O/soak_votes_c3.rc
cat $O/soak_votes_c3.rc
