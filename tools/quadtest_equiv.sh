#!/bin/bash
# tools/quadtest_equiv.sh [N] — host-side equivalence check (no GPU): the register-resident QuadrilateralTest builder of
# csrc/ssd_quadtest.h against the run-time-indexed builder it replaced (taken from commit 00a21d4 of this repository),
# field by field on N random quadrilaterals (treads, arbitrary, many equal coordinates, NaN / inf), every error code included.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
git -C $R show 00a21d4:stair-step-detector_amd/csrc/ssd_quadtest.h | sed -e 's/SSD_QUADTEST_H_/SSD_QUADTEST_OLD_H_/g' \
  -e 's/namespace ssd/namespace ssd_old/' -e 's/__device__ __forceinline__/__host__ __device__ inline/g' \
  -e 's/__device__ inline/__host__ __device__ inline/g' > $T/old_quadtest.h
hipcc -O2 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -x hip $R/tools/quadtest_equiv.cpp -I$T -I$R/stair-step-detector_amd/csrc -I$R/include -o $T/equiv
$T/equiv ${1:-4000000}
rm -rf $T
