#!/bin/bash
# one-off GPU call of round 5 (rewritten per call)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5s; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "tests rc $?" | tee $O/tests.rc; tail -3 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo done
