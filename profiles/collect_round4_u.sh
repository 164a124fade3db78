#!/bin/bash
# Round 4, GPU call u: the default voting arithmetic against the strict one at scale (no oracle: GPU speed) — 1 048 576 C2
# frames (6.3e8 hypotheses) and 65 536 C3 frames (4.8e9).  Same binary as the collection.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4u
mkdir -p $O
cd $R
MPE_SOAK_ORACLE=0 timeout 900 python tests/soak_votes.py 1048576 C2 gpurun_out/r4u/soak_default_vs_strict_C2 > $O/soak_default_vs_strict_C2.log 2>&1; echo "rc $?" >> $O/soak_default_vs_strict_C2.log
MPE_SOAK_ORACLE=0 MPE_SOAK_STRICT_FRAMES=65536 timeout 900 python tests/soak_votes.py 65536 C3 gpurun_out/r4u/soak_default_vs_strict_C3 > $O/soak_default_vs_strict_C3.log 2>&1; echo "rc $?" >> $O/soak_default_vs_strict_C3.log
tail -2 $O/soak_default_vs_strict_C2.log | cut -c1-600; tail -2 $O/soak_default_vs_strict_C3.log | cut -c1-600
