#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for tb in 32768 16384 12288 8192 6144 4096 2048; do
echo -n "target blocks $tb: "
SSD_TARGET_BLOCKS=$tb python bench.py --steps 5 --warmup 2 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value']), {k: round(v,3) for k,v in d['stage_ms'].items()}, round(d['roofline']['frac'],3))"
done
