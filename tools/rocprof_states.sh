#!/bin/bash
# tools/rocprof_states.sh PAD_MIB... — GPU box: the rocprofv3 --kernel-trace --stats pass of the bench command (one batch at a time) once per given pad
# (bench.py --pad-mib: device memory held in front of the frames), K1's average per pass: does a profiled process always draw K1's slow state?
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for pad in "$@"; do
  OUT=$R/gpurun_out/prof_states/pad$pad; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats -d $OUT --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --batches-in-flight 1 --no-cpu --no-hostfed --no-latency --no-secondary --pad-mib $pad > $OUT/log.txt 2>&1
  f=$(ls $OUT/*/*kernel_stats.csv | head -1)
  python3 - "$f" "$pad" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_hist_planes" in r["Name"]:
        ms = float(r["AverageNs"]) / 1e6
        print("pad %5s MiB  k_hist_planes  calls %s  avg %.3f ms  = %.3f of the peak" % (sys.argv[2], r["Calls"], ms, 9663676416.0 / (ms * 1e-3) / 8e12))
PY
done
