#!/bin/bash
# tools/r06_sabotage.sh — GPU box: the tests made for K1's single-precision bands (bin edges, pixel edges, x / y limits) against builds in which a band
# is NOT handed to the doubles (tools/mkvariant.sh sab1 "" -DSSD_SABOTAGE_PRE=1: bin edges; sab2 ... =2: pixel edges).  They must FAIL there and pass on lib/.
R=$GRAFT_REPO_ROOT; cd $R
T="tests/test_gpu_quirks.py::test_points_on_bin_edges_and_pixel_edges_take_the_doubles"
for d in lib lib_sab1 lib_sab2; do
  SSD_HIP_LIB=$R/stair-step-detector_amd/$d/libssd_hip.so timeout -k 10 300 python -m pytest "$T" -q -p no:cacheprovider 2>&1 | tail -4 | sed "s/^/$d: /"
done
