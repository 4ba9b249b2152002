#!/bin/bash
# tools/ab_fhd.sh ROUNDS DIR_A DIR_B .. — bench.py --workload fhd_stress alternately on library builds: frames/s and the stage times
N=$1; shift
for r in $(seq 1 $N); do
  for d in "$@"; do
    SSD_HIP_LIB=$GRAFT_REPO_ROOT/stair-step-detector_amd/$d/libssd_hip.so python bench.py --workload fhd_stress --steps 40 --no-cpu --no-hostfed --no-latency 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$d', round(j['value']), {k: round(v,3) for k,v in j['stage_ms'].items()})"
  done
done
