#!/bin/bash
# tools/exp.sh — quick check on the GPU box: GPU tests, then per-stage ms of the default bench workload
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python bench.py --steps 5 --warmup 2 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value']), {k: round(v,3) for k,v in d['stage_ms'].items()}, round(d['roofline']['frac'],3))"
