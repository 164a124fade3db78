#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3s
rm -rf $O; mkdir -p $O
cd $R && timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "histogram_threshold" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -25 $O/pytest.log
