#!/bin/bash
# tools/r04_gputests.sh — the driver's GPU tier (pytest -m gpu) + smoke, with timing, on the GPU box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_tests; mkdir -p $O
cd $R
( time timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=15 ) > $O/pytest.txt 2>&1
rc=$?
tail -40 $O/pytest.txt
[ $rc -eq 0 ] && python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1 && tail -2 $O/smoke.txt
exit $rc
