#!/usr/bin/env python3
"""tools/cell_analysis.py [N_FRAMES] — CPU-side (oracle + numpy) study of how much of the input the later passes need, on the
frames bench.py times.  Behind the numbers of DESIGN.md section 3: the share of 256-point wave tiles vs 64-point cells that
hold a bin K2 (step plateaus) / K4 (live quadrilaterals) cares about, the lane use inside them, and the share of cells K4
still has to walk once K1's per-cell bounding boxes rule out ground cells wholly outside the ground quadrilateral and
tread cells wholly inside their quadrilateral's constant cell.  TEST INFRASTRUCTURE (uses oracle/); no GPU needed."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
ssd = importlib.import_module("stair-step-detector_amd")
import oracle_binding as ob  # noqa: E402
import scenes  # noqa: E402

oracle = ob.load_oracle()
W, H = 1024, 768
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
sc_list = scenes.batch_scenes(ssd, W, H, n, base_seed=100000, rng_seed=1000)
trans = ssd.transformation_for_scene(sc_list[0])
cfg = ssd.default_config(W, H)
ocfg, ocal = ob.to_oracle_config(cfg), ob.to_oracle_calibration(trans.constants)
A, b = np.array(trans.constants.a).reshape(3, 3), np.array(trans.constants.b)
acc = {k: 0.0 for k in ("k2_tiles", "k2_cells", "k2_use_tiles", "k2_use_cells", "k4_tiles", "k4_cells", "k4_walked", "k4_ground", "k4_tread", "k4_both")}
for sc in sc_list:
    xyz = ssd.synth_host([sc])[0].reshape(-1, 3)
    res, *_ = oracle.process(ocfg, ocal, xyz)
    w = xyz.astype(np.float64) @ A.T + b
    ok = (xyz[:, 2] > 0) & (w[:, 0] > -0.6) & (w[:, 0] < 0.6) & (w[:, 1] > 0.1) & (w[:, 1] < 1.3) & (w[:, 2] > -0.1) & (w[:, 2] < 1.1)
    bins = np.where(ok, ((w[:, 2] + 0.1) * 100.0).astype(np.int64), -1)
    lut, consumed = np.full(128, -1), -1
    for i in range(res.n_plateaus):
        p = res.plateaus[i]
        lo, hi = max(p.bin_lo, consumed + 1), p.bin_hi
        consumed = max(consumed, hi)
        lut[lo:hi + 1] = i
    step = np.array([res.plateaus[i].is_step for i in range(res.n_plateaus)], bool)
    valid = np.array([res.plateaus[i].valid for i in range(res.n_plateaus)], bool)
    g = res.ground_ind
    grp_step, grp_live = np.zeros(32, bool), np.zeros(32, bool)
    quads = {}
    for bb in range(121):
        q = lut[bb]
        if q >= 0 and step[q]:
            grp_step[bb // 4] = True
        if q >= 0 and (q == g or (step[q] and valid[q])):
            grp_live[bb // 4] = True
    for i in range(res.n_plateaus):
        if step[i] and valid[i]:
            quads[i] = np.array(res.plateaus[i].quad_world).reshape(4, 2)
    if g >= 0 and res.first_valid_ind >= 0:
        quads[g] = np.array(res.ground_quad_world).reshape(4, 2)
    pl = np.where(bins >= 0, lut[np.maximum(bins, 0)], -1)
    is_step = (pl >= 0) & step[np.maximum(pl, 0)]
    in_grp_step = (bins >= 0) & grp_step[np.maximum(bins, 0) // 4]
    in_grp_live = (bins >= 0) & grp_live[np.maximum(bins, 0) // 4]
    for unit, key in ((256, "tiles"), (64, "cells")):
        m = in_grp_step.reshape(-1, unit).any(1)
        acc["k2_" + key] += m.mean()
        acc["k2_use_" + key] += is_step.reshape(-1, unit)[m].mean()
        acc["k4_" + key] += in_grp_live.reshape(-1, unit).any(1).mean()
    # K1's boxes on the 256 x 256 grid, K4's classification
    nc = W * H // 64
    okc = ok.reshape(nc, 64)
    qx = np.floor((w[:, 0] + 0.6) * (256 / 1.2)).reshape(nc, 64)
    qy = np.floor((w[:, 1] - 0.1) * (256 / 1.2)).reshape(nc, 64)
    x0, x1 = np.where(okc, qx, 1e9).min(1), np.where(okc, qx, -1).max(1)
    y0, y1 = np.where(okc, qy, 1e9).min(1), np.where(okc, qy, -1).max(1)
    X0, X1 = -0.6 + x0 * (1.2 / 256) - 1e-9, -0.6 + (x1 + 1) * (1.2 / 256) + 1e-9
    Y0, Y1 = 0.1 + y0 * (1.2 / 256) - 1e-9, 0.1 + (y1 + 1) * (1.2 / 256) + 1e-9
    grp = np.where(bins >= 0, bins // 4, -1).reshape(nc, 64)
    need = np.zeros(nc, bool)
    need_g = np.zeros(nc, bool); need_t = np.zeros(nc, bool)
    for i, q in quads.items():
        has = np.zeros(nc, bool)
        for gg in set(bb // 4 for bb in range(128) if lut[bb] == i):
            has |= (grp == gg).any(1)
        xs, ys = np.sort(q[:, 0]), np.sort(q[:, 1])
        if i == g:
            need_g |= has & ~((X1 <= xs[0]) | (X0 >= xs[3]) | (Y1 <= ys[0]) | (Y0 >= ys[3]))
        else:
            need_t |= has & ~((X0 >= xs[1]) & (X1 < xs[2]) & (Y0 >= ys[1]) & (Y1 < ys[2]))
    need = need_g | need_t
    acc["k4_walked"] += need.mean(); acc["k4_ground"] += need_g.mean(); acc["k4_tread"] += need_t.mean(); acc["k4_both"] += (need_g & need_t).mean()
for k in acc:
    acc[k] /= n
print("K2: %.0f %% of the wave tiles at %.0f %% lane use  ->  %.0f %% of the cells at %.0f %%" %
      (100 * acc["k2_tiles"], 100 * acc["k2_use_tiles"], 100 * acc["k2_cells"], 100 * acc["k2_use_cells"]))
print("K4 walked cells: ground %.1f %%, tread %.1f %% (constant cell only; the edge test removes more), both %.1f %%" % (100 * acc["k4_ground"], 100 * acc["k4_tread"], 100 * acc["k4_both"]))
print("K4: %.0f %% of the wave tiles  ->  %.0f %% of the cells  ->  %.0f %% with the boxes" %
      (100 * acc["k4_tiles"], 100 * acc["k4_cells"], 100 * acc["k4_walked"]))
