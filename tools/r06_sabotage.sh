#!/bin/bash
# tools/r06_sabotage.sh — GPU box: the tests made for the single-precision bands (K1 and k_inquad: bin edges, pixel edges, x / y limits; k_inquad: the
# edges of the quadrilaterals) against builds in which a band is NOT handed to the doubles (tools/mkvariant.sh sab1 "" -DSSD_SABOTAGE_PRE=1: bin edges;
# sab2 ... =2: pixel edges; sab4 ... =4: quadrilateral edges).  The band's test must FAIL there; both pass on lib/.
R=$GRAFT_REPO_ROOT; cd $R
T1="tests/test_gpu_quirks.py::test_points_on_bin_edges_and_pixel_edges_take_the_doubles"
T2="tests/test_gpu_quirks.py::test_points_on_the_edges_of_the_quadrilaterals_take_the_doubles"
for d in lib lib_sab1 lib_sab2 lib_sab4; do
  SSD_HIP_LIB=$R/stair-step-detector_amd/$d/libssd_hip.so timeout -k 10 300 python -m pytest "$T1" "$T2" -q -p no:cacheprovider 2>&1 | grep -v "^$\|^=\+ FAILURES\|^E \|^tests/\|^    \|^_\+ \|^[a-z_]* = " | tail -8 | sed "s/^/$d: /"
done
