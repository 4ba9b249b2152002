#!/bin/bash
# tools/sample_ab.sh ROUNDS LIBDIR... — bench.py (three batches in flight, 24 steps) alternately on library builds: value, K1, predict, raster, frames covered
N=$1; shift
for r in $(seq 1 $N); do for d in "$@"; do
  SSD_HIP_LIB=$GRAFT_REPO_ROOT/stair-step-detector_amd/$d/libssd_hip.so python3 bench.py --no-cpu --no-hostfed --no-latency --no-secondary --steps 24 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
sp=d['single_pass']
print('%-10s value %8.0f  ms/step %.3f  K1 %.3f  predict %.3f  raster %.3f  covered %s of %s  planes/frame %.2f' % ('$d', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['predict_ms'], d['stage_ms']['raster'], sp['frames_covered_by_the_predictor'], sp['frames_with_steps'], sp['planes_per_frame']))"
done; done
