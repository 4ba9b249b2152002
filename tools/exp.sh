#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value']), d['ms_per_step'], sum(d['stage_ms'].values()), {k: round(v,3) for k,v in d['stage_ms'].items()})"
