#!/bin/bash
# tools/r05_depths.sh ROUNDS — bench.py's value by ssd_config::batches_in_flight, alternating, XGA (GPU box)
for r in $(seq 1 ${1:-2}); do
  for d in 3 4 5 6 8; do
    python3 bench.py --batches-in-flight $d --steps 24 --no-cpu --no-hostfed --no-latency --no-secondary 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('in flight %d  value %8.0f  ms/step %.3f' % ($d, d['value'], d['ms_per_step']))"
  done
done
