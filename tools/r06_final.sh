#!/bin/bash
# tools/r06_final.sh TAG — the round's evidence on the GPU box, from ONE build: GPU tests, counters, traces, bench lines, a fuzz
# sweep.  Run tools/stamp.py in the build container first: the box checks that the library it runs is the stamped one and writes
# the stamp (git revision of the build, sha256 of libssd_hip.so) next to everything it collects; tools/parse_profiles.py copies it
# into the files kept under profiles/.
TAG=${1:-r06_final}; R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/$TAG gpurun_out/prof_$TAG
python3 - <<PY || exit 1
import hashlib, json, sys
st = json.load(open("build/STAMP.json"))
h = hashlib.sha256(open("stair-step-detector_amd/lib/libssd_hip.so", "rb").read()).hexdigest()
if h != st["lib_sha256"]:
    sys.exit("tools/r06_final.sh: lib/libssd_hip.so (%s) is not the stamped build (%s): run tools/stamp.py after the last make" % (h[:16], st["lib_sha256"][:16]))
st["checked_on_gpu_box"] = True
json.dump(st, open("gpurun_out/prof_$TAG/stamp.json", "w"), indent=1)
print("stamp", json.dumps(st))
PY
( time timeout -k 10 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/$TAG/pytest.txt 2>&1 || { tail -20 gpurun_out/$TAG/pytest.txt; exit 1; }
tail -4 gpurun_out/$TAG/pytest.txt
PMC_BENCH_ARGS="--batches-in-flight 1" bash tools/pmc.sh $TAG "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "VALUBusy SALUBusy" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "MemUnitStalled LDSBankConflict" > gpurun_out/$TAG/pmc.txt 2>&1
bash tools/profile.sh $TAG > gpurun_out/$TAG/profile.txt 2>&1
python3 tools/fuzz.py ${FUZZ_POSES:-100} 384 ${FUZZ_SEED:-60603} mixed > gpurun_out/$TAG/fuzz.txt 2>&1; tail -1 gpurun_out/$TAG/fuzz.txt | cut -c1-400
python3 tools/fuzz_slabs.py > gpurun_out/$TAG/fuzz_slabs.txt 2>&1; tail -1 gpurun_out/$TAG/fuzz_slabs.txt | cut -c1-300
# a GPU fault anywhere above fails the run, whatever the steps' exit codes were
if grep -l -a "Memory access fault\|GPU core dump" gpurun_out/$TAG/*.txt gpurun_out/prof_$TAG/*.log gpurun_out/pmc_$TAG/*.log 2>/dev/null; then echo "GPU fault reported in the files above"; exit 1; fi
