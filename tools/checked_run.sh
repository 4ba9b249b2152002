#!/bin/bash
# tools/checked_run.sh [POSES] — on the GPU box: the whole -m gpu tier, tools/fuzz.py (mixed, then with the predictor sabotaged) and
# tools/fuzz_slabs.py against the bounds-checked build (tools/checked.sh).  Every violation prints "SSD_CHECK site ..."; the summary
# line at the end counts them.  Output: gpurun_out/checked/*.txt
R=$GRAFT_REPO_ROOT; cd $R
export SSD_HIP_LIB=$R/stair-step-detector_amd/lib_checked/libssd_hip.so
N=${1:-130}
mkdir -p gpurun_out/checked
timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_capi.py > gpurun_out/checked/pytest.txt 2>&1; tail -3 gpurun_out/checked/pytest.txt
python3 tools/fuzz.py $N 384 424242 mixed > gpurun_out/checked/fuzz_mixed.txt 2>&1; tail -1 gpurun_out/checked/fuzz_mixed.txt | cut -c1-300
FUZZ_SABOTAGE=1 python3 tools/fuzz.py $((N / 2)) 384 434343 mixed > gpurun_out/checked/fuzz_sabotage.txt 2>&1; tail -1 gpurun_out/checked/fuzz_sabotage.txt | cut -c1-300
python3 tools/fuzz_slabs.py > gpurun_out/checked/fuzz_slabs.txt 2>&1; tail -1 gpurun_out/checked/fuzz_slabs.txt | cut -c1-300
echo "SSD_CHECK reports: $(cat gpurun_out/checked/*.txt | grep -a -c 'SSD_CHECK site')   GPU faults: $(cat gpurun_out/checked/*.txt | grep -a -c 'Memory access fault')"
