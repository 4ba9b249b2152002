#!/bin/bash
# tools/r04_ab.sh DIR_A DIR_B .. — GPU tests on the product build, then alternating A/B of library variants: throughput with three
# batches in flight (tools/ab_inflight.py) and per-stage times one batch at a time (tools/ab.sh), XGA
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 || exit 1
python tools/ab_inflight.py "$@"
bash tools/ab.sh 2 "$@"
