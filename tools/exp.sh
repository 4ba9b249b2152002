#!/bin/bash
# tools/exp.sh TAG "ENV=VAL ENV=VAL" ... — one short bench per environment setting (GPU box, via gpurun); prints the stage
# times.  A setting that contains the word FHD runs the fhd_stress workload.
# The SSD_* geometry variables are read only by the tools build of the library (the product has no getenv):
#   make -C stair-step-detector_amd/csrc OUT=../lib_tuning EXTRA=-DSSD_TUNING ../lib_tuning/libssd_hip.so
export SSD_HIP_LIB=${SSD_HIP_LIB:-$GRAFT_REPO_ROOT/stair-step-detector_amd/lib_tuning/libssd_hip.so}
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
i=0
for envs in "$@"; do
  i=$((i+1))
  wl="xga_batch"; case "$envs" in *FHD*) wl="fhd_stress";; esac
  env $(echo $envs | sed 's/FHD//') timeout 200 python bench.py --workload $wl --steps 8 --warmup 2 --no-cpu --no-secondary > $OUT/b$i.json 2> $OUT/b$i.err
  python - "$OUT/b$i.json" "$envs" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("%-40s %8.0f f/s  %s" % (sys.argv[2], d["value"], {k: round(v, 3) for k, v in d["stage_ms"].items()}))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
