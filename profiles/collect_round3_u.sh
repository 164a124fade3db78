#!/bin/bash
# final check: smoke() and the whole GPU suite on the committed tree
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3u
rm -rf $O; mkdir -p $O
cd $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -3 $O/smoke.log; tail -3 $O/pytest.log
