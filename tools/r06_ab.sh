#!/bin/bash
# tools/r06_ab.sh TAG LIBDIR... — GPU box: tools/stages.py alternately on the named library builds (XGA, then FHD stress), then one PMC pass
# (vector / scalar / LDS instructions per kernel) on each of them; into gpurun_out/r06_TAG_*
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd $R
bash tools/ab.sh ${AB_ROUNDS:-3} "$@" > gpurun_out/r06_${TAG}_ab.txt 2>&1
cat gpurun_out/r06_${TAG}_ab.txt
if [ -z "$NO_FHD" ]; then
  STAGES_FHD=1 AB_FRAMES=256 bash tools/ab.sh ${AB_ROUNDS:-3} "$@" > gpurun_out/r06_${TAG}_ab_fhd.txt 2>&1
  cat gpurun_out/r06_${TAG}_ab_fhd.txt
fi
for d in ${PMC_LIBS}; do
  SSD_HIP_LIB=$R/stair-step-detector_amd/$d/libssd_hip.so bash tools/pmc.sh ${TAG}_$d "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" 2>&1 | grep "k_hist\|k_inquad" | sed "s/^/$d /"
done
