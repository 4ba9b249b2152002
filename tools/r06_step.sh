#!/bin/bash
# tools/r06_step.sh TAG [LIBDIR...] — GPU box: the whole -m gpu tier on lib/, then (unless the tier was killed) tools/stages.py alternately on the
# named library builds, XGA and FHD stress; everything into gpurun_out/r06_TAG_*.txt
TAG=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 ${TESTS_TIMEOUT:-500} python -m pytest tests -m gpu -x -q ${PYTEST_ARGS} > gpurun_out/r06_${TAG}_tests.txt 2>&1
rc=$?
tail -3 gpurun_out/r06_${TAG}_tests.txt
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests killed: no timing"; exit $rc; fi
if [ $# -gt 0 ]; then
  bash tools/ab.sh ${AB_ROUNDS:-2} "$@" > gpurun_out/r06_${TAG}_ab.txt 2>&1
  cat gpurun_out/r06_${TAG}_ab.txt
  STAGES_FHD=1 AB_FRAMES=256 bash tools/ab.sh ${AB_ROUNDS:-2} "$@" > gpurun_out/r06_${TAG}_ab_fhd.txt 2>&1
  cat gpurun_out/r06_${TAG}_ab_fhd.txt
fi
exit $rc
