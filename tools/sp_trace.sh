#!/bin/bash
# tools/sp_trace.sh [LIBDIR] — kernel trace of the default bench, one batch in flight (GPU box, via gpurun): per-kernel average durations
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/sp_trace; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ -n "$1" ]; then export SSD_HIP_LIB=$R/stair-step-detector_amd/$1/libssd_hip.so; fi
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/trace --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --batches-in-flight 1 --no-cpu --no-hostfed --no-latency --no-secondary > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
cd $R
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-60s calls %5s avg %10.1f us  %5.1f %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
