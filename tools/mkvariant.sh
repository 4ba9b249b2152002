#!/bin/bash
# tools/mkvariant.sh NAME [GIT_REV] [EXTRA_FLAGS] [SED_EXPR] — builds stair-step-detector_amd/lib_NAME/libssd_hip.so for A/B runs (tools/ab.sh,
# tools/ab_inflight.py): the current tree, with csrc/ssd_kernels.hip taken from GIT_REV when given (the kernels' launch interface
# must still match the current ssd_launch.h).  SED_EXPR, when given, is applied to the copy of ssd_kernels.hip
# (experiments that are not worth a macro in the source).  Tools only; lib_* is git-ignored and travels to the GPU box.
set -e
NAME=$1; REV=$2; EXTRA=$3; SEDX=$4
R=$(cd $(dirname $0)/.. && pwd)
T=$(mktemp -d)
cp -r $R/stair-step-detector_amd/csrc $T/csrc
mkdir -p $T/include && cp -r $R/include/* $T/include/
mkdir -p $T/x && mv $T/csrc $T/x/csrc && mkdir -p $T/pkg && mv $T/x/csrc $T/pkg/csrc
# the Makefile reaches the headers through ../../include
if [ -n "$REV" ]; then git -C $R show $REV:stair-step-detector_amd/csrc/ssd_kernels.hip > $T/pkg/csrc/ssd_kernels.hip; fi
if [ -n "$SEDX" ]; then sed -i -e "$SEDX" $T/pkg/csrc/ssd_kernels.hip; fi
mkdir -p $R/stair-step-detector_amd/lib_$NAME
make -C $T/pkg/csrc OUT=$R/stair-step-detector_amd/lib_$NAME EXTRA="$EXTRA" 2>&1 | grep -E "error|Error" || true
ls -la $R/stair-step-detector_amd/lib_$NAME/libssd_hip.so
rm -rf $T
