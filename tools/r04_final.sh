#!/bin/bash
# tools/r04_final.sh TAG — the round's evidence on the GPU box: GPU tests, counters, traces, bench lines, a fuzz sweep
TAG=${1:-r04_b}; R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/$TAG
( time timeout -k 10 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/$TAG/pytest.txt 2>&1 || { tail -20 gpurun_out/$TAG/pytest.txt; exit 1; }
tail -4 gpurun_out/$TAG/pytest.txt
PMC_BENCH_ARGS="--batches-in-flight 1" bash tools/pmc.sh $TAG "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "VALUBusy SALUBusy" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "MemUnitStalled LDSBankConflict" > gpurun_out/$TAG/pmc.txt 2>&1
bash tools/profile.sh $TAG > gpurun_out/$TAG/profile.txt 2>&1
python3 tools/fuzz.py 176 384 20264 mixed > gpurun_out/$TAG/fuzz.txt 2>&1; tail -1 gpurun_out/$TAG/fuzz.txt | cut -c1-400
python3 tools/fuzz_slabs.py > gpurun_out/$TAG/fuzz_slabs.txt 2>&1; tail -1 gpurun_out/$TAG/fuzz_slabs.txt | cut -c1-300
# a GPU fault anywhere above fails the run, whatever the steps' exit codes were
if grep -l -a "Memory access fault\|GPU core dump" gpurun_out/$TAG/*.txt gpurun_out/prof_$TAG/*.log gpurun_out/pmc_$TAG/*.log 2>/dev/null; then echo "GPU fault reported in the files above"; exit 1; fi
