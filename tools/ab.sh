#!/bin/bash
# tools/ab.sh ROUNDS DIR_A DIR_B [DIR_C ..] — tools/stages.py alternately on the library builds stair-step-detector_amd/DIR_*
# (same box, interleaved: the GPU's clock state drifts by several per cent within a minute)
N=$1; shift
for r in $(seq 1 $N); do
  for d in "$@"; do
    SSD_HIP_LIB=$GRAFT_REPO_ROOT/stair-step-detector_amd/$d/libssd_hip.so STAGES_TAG="$d" python tools/stages.py ${AB_FRAMES:-1024} 8
  done
done
