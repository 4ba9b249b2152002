#!/bin/bash
# tools/exp_frames.sh TAG "FRAMES ENV=VAL ..." ... — like tools/exp.sh with --frames as the first word of each setting
# The SSD_* geometry variables are read only by the tools build of the library (the product has no getenv):
#   make -C stair-step-detector_amd/csrc OUT=../lib_tuning EXTRA=-DSSD_TUNING ../lib_tuning/libssd_hip.so
export SSD_HIP_LIB=${SSD_HIP_LIB:-$GRAFT_REPO_ROOT/stair-step-detector_amd/lib_tuning/libssd_hip.so}
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
i=0
for spec in "$@"; do
  i=$((i+1))
  f=${spec%% *}; envs=${spec#* }; [ "$envs" = "$spec" ] && envs="A=1"
  env $envs timeout 200 python bench.py --frames $f --steps 20 --warmup 3 --no-cpu --no-hostfed --no-latency --no-secondary > $OUT/b$i.json 2> $OUT/b$i.err
  python - "$OUT/b$i.json" "$spec" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("%-44s %8.0f f/s  %s" % (sys.argv[2], d["value"], {k: round(v, 4) for k, v in d["stage_ms"].items()}))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
