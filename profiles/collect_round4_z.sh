#!/bin/bash
# Round 4, GPU call z: longer soaks against the oracle with the collection's binary — end to end 1 048 576 C2 frames,
# vote histograms (three arithmetics) on 16 384 C3 frames.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4z
mkdir -p $O
cd $R
timeout 1200 python tests/soak_parity.py 1048576 C2 65536 gpurun_out/r4z/soak_parity_C2_1m > $O/soak_parity_C2_1m.log 2>&1; echo "rc $?" >> $O/soak_parity_C2_1m.log
MPE_SOAK_STRICT_FRAMES=16384 timeout 1500 python tests/soak_votes.py 16384 C3 gpurun_out/r4z/soak_votes_C3_16k > $O/soak_votes_C3_16k.log 2>&1; echo "rc $?" >> $O/soak_votes_C3_16k.log
tail -2 $O/soak_parity_C2_1m.log | cut -c1-700; tail -2 $O/soak_votes_C3_16k.log | cut -c1-900
