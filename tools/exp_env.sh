#!/bin/bash
# tools/exp_env.sh "ENV=VAL ..." ... — tools/stages.py once per environment setting on the -DSSD_TUNING build (GPU box)
export SSD_HIP_LIB=$GRAFT_REPO_ROOT/stair-step-detector_amd/lib_tuning/libssd_hip.so
for envs in "$@"; do
  env $envs STAGES_TAG="$envs" python tools/stages.py 1024 10
done
