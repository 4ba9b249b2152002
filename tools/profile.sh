#!/bin/bash
# tools/profile.sh TAG — run on the GPU box (via gpurun): kernel trace + stats of the bench command, then
# HBM traffic counters in separate passes (FETCH_SIZE, WRITE_SIZE: MI355X_MICROARCH.md "HBM" / "rocprofv3 PMC slots").
# The traced passes run with --batches-in-flight 1: the default handle keeps three batches in flight, whose kernels share
# the GPU, and a traced duration would be a kernel's share of the machine, not its own time (bench.py takes its roofline
# from separate one-at-a-time steps for the same reason).
# Outputs under gpurun_out/prof_TAG/; tools/parse_profiles.py turns them into the summaries kept in profiles/.
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --batches-in-flight 1 --no-cpu --no-hostfed --no-latency --no-secondary > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --batches-in-flight 1 --no-cpu --no-hostfed --no-latency --no-secondary > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --batches-in-flight 1 --no-cpu --no-hostfed --no-latency --no-secondary > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/trace_fhd --output-format csv -- python3 $R/bench.py --workload fhd_stress --steps 5 --warmup 2 --batches-in-flight 1 --no-cpu --no-latency --no-secondary > $OUT/trace_fhd.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch_fhd --output-format csv -- python3 $R/bench.py --workload fhd_stress --steps 3 --warmup 1 --batches-in-flight 1 --no-cpu --no-latency --no-secondary > $OUT/fetch_fhd.log 2>&1
cd $R
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 bench.py --workload fhd_stress --cpu-frames 48 > $OUT/bench_fhd.json 2> $OUT/bench_fhd.err
python3 bench.py --input depth16 --no-hostfed --cpu-frames 64 > $OUT/bench_depth16.json 2> $OUT/bench_depth16.err
python3 tools/hostfed.py 256 > $OUT/hostfed.json 2>&1
python3 tools/latency.py > $OUT/latency.txt 2>&1
ls -R $OUT | head -50
