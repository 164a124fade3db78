#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3s
rm -rf $O; mkdir -p $O
cd $R && timeout 600 python -m pytest tests/test_gpu_parity_large.py -m gpu -x -q -k "vote_events or streaming" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -15 $O/pytest.log
