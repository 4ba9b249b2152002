#!/bin/bash
# tools/r04_step.sh TAG [extra] — one build's check on the GPU box: the GPU tier of the tests, then the XGA and FHD bench lines
TAG=${1:-step}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_$TAG; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
python3 bench.py --no-cpu --no-hostfed --no-latency > $O/bench.json 2> $O/bench.err || { tail $O/bench.err; exit 1; }
python3 bench.py --workload fhd_stress --no-cpu --no-hostfed --no-latency > $O/bench_fhd.json 2> $O/bench_fhd.err
python3 bench.py --input depth16 --no-cpu --no-hostfed --no-latency > $O/bench_d16.json 2> $O/bench_d16.err
python3 - $O <<'PY'
import json, sys
for f in ("bench", "bench_fhd", "bench_d16"):
    try:
        d = json.load(open("%s/%s.json" % (sys.argv[1], f)))
        print("%-10s %8.0f f/s  one-at-a-time %.3f ms  %s  k1/stream %s" % (f, d["value"], d["one_batch_at_a_time"]["ms_per_step"], {k: round(v, 3) for k, v in d["stage_ms"].items()}, d["roofline"]["k1_over_plain_stream"]))
    except Exception as e:
        print(f, "FAILED", e)
PY
if [ "$2" = "rates" ]; then hipcc --offload-arch=gfx950 -O3 tools/instr_rate.hip -o /tmp/instr_rate && /tmp/instr_rate | tee $O/instr_rate.txt; fi
