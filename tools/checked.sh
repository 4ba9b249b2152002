#!/bin/bash
# tools/checked.sh — builds stair-step-detector_amd/lib_checked/ with -DSSD_CHECKED (ssd_kernels.hip: every store or atomic whose
# address comes from a point, a pixel, a window or a list is bounds-checked on the device first and reported instead of performed).
# Tools only, never shipped.  On the GPU box: tools/checked_run.sh runs the GPU tier and the fuzz sweeps against it.
set -e
R=$(cd $(dirname $0)/.. && pwd)
bash $R/tools/mkvariant.sh checked "" "-DSSD_CHECKED"
