#!/bin/bash
# tools/lat.sh TAG — kernel trace of single-frame calls (GPU box)
TAG=${1:-lat}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/t1 --output-format csv -- python3 $R/tools/latency_trace.py 1 > $OUT/t1.log 2>&1
cd $R
python3 tools/latency_gaps.py $OUT/t1 | tee $OUT/gaps1.txt
python3 tools/latency.py | tee $OUT/latency.txt
