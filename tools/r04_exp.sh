#!/bin/bash
# tools/r04_exp.sh DIR_A DIR_B .. — alternating A/B of library variants without the test tier (experiments whose results are
# not expected to be right): three batches in flight, then per-stage times
R=$GRAFT_REPO_ROOT; cd $R
python tools/ab_inflight.py "$@" && bash tools/ab.sh ${AB_ROUNDS:-1} "$@"
