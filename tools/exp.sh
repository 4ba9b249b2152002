#!/bin/bash
# quick A/B on the GPU box: tests, then the XGA and FHD benches (extra bench.py arguments may be given)
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
show='import json,sys; d=json.load(sys.stdin); print(round(d["value"]), round(d["ms_per_step"],3), round(sum(d["stage_ms"].values()),3), {k: round(v,3) for k,v in d["stage_ms"].items()})'
python bench.py --steps 10 --warmup 2 --no-cpu "$@" 2>/dev/null | python -c "$show"
python bench.py --workload fhd_stress --steps 10 --warmup 2 --no-cpu "$@" 2>/dev/null | python -c "$show"
