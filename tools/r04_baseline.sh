#!/bin/bash
# tools/r04_baseline.sh — round-4 opening measurements on the GPU box: bench lines, block phases, chunk sweep with three batches in flight
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_base; mkdir -p $O
cd $R
python3 bench.py --no-cpu --no-hostfed --no-latency > $O/bench.json 2> $O/bench.err && echo bench done
python3 bench.py --workload fhd_stress --no-cpu --no-hostfed --no-latency > $O/bench_fhd.json 2> $O/bench_fhd.err && echo fhd done
python3 tools/blockphases.py 1024 > $O/blockphases.txt 2>&1 && echo phases done
bash tools/exp.sh r04_base/chunks "SSD_K2_CHUNK_TILES=32 SSD_K4_CHUNK_TILES=16" "SSD_K2_CHUNK_TILES=64 SSD_K4_CHUNK_TILES=16" "SSD_K2_CHUNK_TILES=32 SSD_K4_CHUNK_TILES=32" "SSD_K2_CHUNK_TILES=64 SSD_K4_CHUNK_TILES=32" "SSD_K2_CHUNK_TILES=128 SSD_K4_CHUNK_TILES=64" "SSD_K2_CHUNK_TILES=16 SSD_K4_CHUNK_TILES=8" "SSD_K2_CHUNK_TILES=32 SSD_K4_CHUNK_TILES=16" > $O/chunks.txt 2>&1
cat $O/chunks.txt
