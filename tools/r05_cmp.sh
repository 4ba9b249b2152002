#!/bin/bash
# tools/r05_cmp.sh LIBDIR... — bench.py's value / K1 time for XGA, FHD stress and depth-16 input on each library build (GPU box)
for d in "$@"; do
  export SSD_HIP_LIB=$GRAFT_REPO_ROOT/stair-step-detector_amd/$d/libssd_hip.so
  for w in "" "--workload fhd_stress" "--input depth16"; do
    python3 bench.py $w --no-cpu --no-hostfed --no-latency --no-secondary 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-10s %-24s value %8.0f  ms/step %.3f  K1 %.3f  frac %.3f  one-at-a-time %.3f  stages %s' % ('$d', '$w', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['one_batch_at_a_time']['ms_per_step'], ' '.join('%s %.3f' % (k[:4], v) for k, v in d['stage_ms'].items())))"
  done
done
