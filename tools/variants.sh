#!/bin/bash
# tools/variants.sh TAG DIR... — for every library directory (stair-step-detector_amd/DIR, a build variant): single-frame
# latency and a short XGA batch bench (GPU box)
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for d in "$@"; do
  export SSD_HIP_LIB=$GRAFT_REPO_ROOT/stair-step-detector_amd/$d/libssd_hip.so
  echo "== $d"
  python tools/latency.py 2>&1 | cut -c1-260 | tee $OUT/lat_$d.txt
  timeout 200 python bench.py --steps 8 --warmup 2 --no-cpu --no-hostfed --no-latency > $OUT/b_$d.json 2> $OUT/b_$d.err
  python - "$OUT/b_$d.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("   batch %8.0f f/s  %s  pipelined %s" % (d["value"], {k: round(v, 3) for k, v in d["stage_ms"].items()}, {k: round(v["value"]) for k, v in d.get("pipelined", {}).items() if isinstance(v, dict)}))
PY
done
