#!/bin/bash
# tools/pmc.sh OUTTAG "COUNTER COUNTER ..." — one rocprofv3 --pmc pass of the default bench (GPU box, via gpurun)
TAG=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "$@"; do
  name=$(echo $set | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $set -d $OUT/$name --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-hostfed --no-latency --no-secondary ${PMC_BENCH_ARGS} > $OUT/$name.log 2>&1
  python3 - "$OUT/$name" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")
if not f:
    print("no counters for", sys.argv[1]); sys.exit(0)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if k.startswith("ssd::k_") and "synth" not in k:
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k, {c: round(sum(v) / len(v), 3) for c, v in agg[k].items()})
PY
done
