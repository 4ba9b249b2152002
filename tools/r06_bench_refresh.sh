#!/bin/bash
# tools/r06_bench_refresh.sh TAG — on the GPU box, AFTER the counters of tools/r06_final.sh have been parsed into profiles/ and
# committed: the three bench lines again, so that the lines kept under profiles/ cite the counters of their own build (the lines
# r06_final.sh wrote were made while profiles/ still held the build before: `stale`).  Refuses another library than the stamped one.
TAG=${1:-r06_final}; R=$GRAFT_REPO_ROOT; cd $R
OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT
python3 - <<PY || exit 1
import hashlib, json, sys
st = json.load(open("build/STAMP.json"))
h = hashlib.sha256(open("stair-step-detector_amd/lib/libssd_hip.so", "rb").read()).hexdigest()
if h != st["lib_sha256"]:
    sys.exit("lib/libssd_hip.so is not the stamped build")
PY
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err && tail -c 300 $OUT/bench.json
python3 bench.py --workload fhd_stress --cpu-frames 48 > $OUT/bench_fhd.json 2> $OUT/bench_fhd.err
python3 bench.py --input depth16 --no-hostfed --cpu-frames 64 > $OUT/bench_depth16.json 2> $OUT/bench_depth16.err
python3 bench.py --no-cpu --no-hostfed --no-latency --no-secondary --steps 24 > $OUT/bench_steps24.json 2> $OUT/bench_steps24.err
grep -o '"stale": [a-z]*' $OUT/bench.json $OUT/bench_fhd.json $OUT/bench_depth16.json | sort | uniq -c
