#!/bin/bash
# tools/exp.sh — timing experiments on the GPU box (not product code): per-stage ms under env switches
cd $GRAFT_REPO_ROOT
for cfg in "SSD_EXP=0" $EXTRA; do
  echo "== $cfg"
  env $cfg python bench.py --steps 4 --warmup 2 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value']), {k: round(v,3) for k,v in d['stage_ms'].items()})"
done
