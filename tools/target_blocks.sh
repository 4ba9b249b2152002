#!/bin/bash
# tools/target_blocks.sh — sweep of the block size (points per block) of the streaming kernels, XGA batch and FHD stress
# The SSD_* geometry variables are read only by the tools build of the library (the product has no getenv):
#   make -C stair-step-detector_amd/csrc OUT=../lib_tuning EXTRA=-DSSD_TUNING ../lib_tuning/libssd_hip.so
export SSD_HIP_LIB=${SSD_HIP_LIB:-$GRAFT_REPO_ROOT/stair-step-detector_amd/lib_tuning/libssd_hip.so}
cd $GRAFT_REPO_ROOT
show='import json,sys; d=json.load(sys.stdin); print(round(d["value"]), round(d["ms_per_step"],3), {k: round(v,3) for k,v in d["stage_ms"].items()})'
for cp in 16384 24576 32768 49152 65536; do
  echo "== SSD_CHUNK_POINTS=$cp"
  SSD_CHUNK_POINTS=$cp python bench.py --steps 10 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python -c "$show"
  SSD_CHUNK_POINTS=$cp python bench.py --workload fhd_stress --steps 10 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python -c "$show"
done
