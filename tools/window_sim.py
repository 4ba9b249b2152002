#!/usr/bin/env python3
"""tools/window_sim.py [N_FRAMES] — CPU model of k_raster's per-wave LDS write-combining window on the frames bench.py times:
words that hit the window, words that miss it (each a global atomic), flushes and words flushed, for several window
shapes and re-anchoring rules.  Behind the choice in DESIGN.md section 3.  TEST INFRASTRUCTURE (uses oracle/); no GPU needed."""
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

W, H = 1024, 768
W64 = W // 64
CELL, COLS = 64, 16
CHUNK_TILES = 32                      # k_raster's chunk at 1024 frames (t2)
WAVES = 4


def frame_points(oracle, ocfg, ocal, A, b, sc):
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
    pl = np.where(bins >= 0, lut[np.maximum(bins, 0)], -1)
    slot = np.where((pl >= 0) & step[np.maximum(pl, 0)], pl, -1)
    grp_step = np.zeros(32, bool)
    for bb in range(121):
        if lut[bb] >= 0 and step[lut[bb]]:
            grp_step[bb // 4] = True
    want_pt = (bins >= 0) & grp_step[np.maximum(bins, 0) // 4]
    ix = ((w[:, 0] + 0.6) * (W / 1.2)).astype(np.int64)
    iy = ((1.3 - w[:, 1]) * (H / 1.2)).astype(np.int64)
    return slot, ix, iy, want_pt.reshape(-1, CELL).any(1)


def simulate(slot, ix, iy, want_cell, win_rows, win_cols, rule):
    """-> (emitted words, misses, flushes, flushed words)"""
    n_cells = len(want_cell)
    per_chunk = CHUNK_TILES * 16
    emitted = misses = flushes = flushed = 0
    slot = slot.reshape(n_cells, 16, 4); ix = ix.reshape(n_cells, 16, 4); iy = iy.reshape(n_cells, 16, 4)
    for c0 in range(0, n_cells, per_chunk):
        nc = min(per_chunk, n_cells - c0)
        rows = (nc + COLS - 1) // COLS
        order = []
        for cx in range(COLS):
            col = [r * COLS + cx for r in range(rows) if r * COLS + cx < nc and want_cell[c0 + r * COLS + cx]]
            if "pad" in rule and len(col) % 4:
                col += [-1] * (4 - len(col) % 4)
            order += col
        if "rowwin" in rule:
            e, m, f, fw = simulate_rowwin(slot, ix, iy, c0, order, win_rows, win_cols)
            emitted += e; misses += m; flushes += f; flushed += fw
            continue
        n_groups = (len(order) + 3) // 4
        for wv in range(WAVES):
            g0, g1 = wv * n_groups // WAVES, (wv + 1) * n_groups // WAVES
            wslot, row0, col0 = -1, 0, 0
            touched = set()
            for g in range(g0, g1):
                cells = order[4 * g:4 * g + 4]
                em_lanes = 0
                miss_keys = []
                all_keys = []
                lane_first = []
                for c in cells:
                    if c < 0:
                        continue
                    s4, x4, y4 = slot[c0 + c], ix[c0 + c], iy[c0 + c]
                    for ln in range(16):
                        pend = None
                        lane_em = False
                        for j in range(4):
                            s = s4[ln, j]
                            if s < 0:
                                continue
                            key = (int(s), int(y4[ln, j]), int(x4[ln, j]) >> 6)
                            if rule.startswith("pt"):
                                all_keys.append(key); lane_em = True
                                continue
                            if pend is not None and key != pend:
                                all_keys.append(pend); lane_em = True
                            pend = key
                        if pend is not None:
                            all_keys.append(pend); lane_em = True
                        em_lanes += lane_em
                        lane_keys = [(int(s4[ln, j]), int(y4[ln, j]), int(x4[ln, j]) >> 6) for j in range(4) if s4[ln, j] >= 0]
                        if lane_keys:
                            lane_first.append(min(lane_keys))
                if rule.startswith("pre") and lane_first:
                    def inwin(k):
                        return k[0] == wslot and 0 <= k[1] - row0 < win_rows and 0 <= k[2] - col0 < win_cols
                    missing = [k for k in lane_first if not inwin(k)]
                    if 2 * len(missing) > len(lane_first):
                        if wslot >= 0:
                            flushes += 1; flushed += len(touched)
                        touched = set()
                        lowest = min(missing)
                        wslot, row0 = lowest[0], lowest[1]
                        col0 = max(0, min(lowest[2] - 1, W64 - win_cols))
                for key in all_keys:
                    emitted += 1
                    s, y, x = key
                    if s == wslot and 0 <= y - row0 < win_rows and 0 <= x - col0 < win_cols:
                        touched.add((y, x))
                    else:
                        misses += 1
                        miss_keys.append(key)
                if not miss_keys or rule.startswith("pre"):
                    continue
                lowest = min(miss_keys)
                if rule.startswith("half"):
                    re = 2 * len(miss_keys) > em_lanes
                elif rule == "pthalf":
                    re = 2 * len(miss_keys) > len(all_keys)
                elif rule == "ptquarter":
                    re = 4 * len(miss_keys) > len(all_keys)
                elif rule == "half":
                    re = 2 * len(miss_keys) > em_lanes
                elif rule == "any4":
                    re = len(miss_keys) >= 4 and (2 * len(miss_keys) > em_lanes or all(k[0] == wslot for k in miss_keys))
                elif rule == "quarter":
                    re = 4 * len(miss_keys) > em_lanes
                if re:
                    if wslot >= 0:
                        flushes += 1; flushed += len(touched)
                    touched = set()
                    wslot, row0 = lowest[0], lowest[1]
                    if rule == "any4":
                        # anchor at the lowest row touched in this tile by that slot (the walk goes down the image)
                        row0 = min(k[1] for k in all_keys if k[0] == wslot)
                    col0 = max(0, min(lowest[2] - (1 if win_cols >= 4 else 0), W64 - win_cols))
            if wslot >= 0:
                flushes += 1; flushed += len(touched)
    return emitted, misses, flushes, flushed


def simulate_rowwin(slot, ix, iy, c0, order, win_rows, win_cols):
    """every DPP row (16 lanes = one cell per iteration) has its own window and walks a contiguous part of the list"""
    emitted = misses = flushes = flushed = 0
    n = len(order)
    parts = WAVES * 4
    for p in range(parts):
        a, b = p * n // parts, (p + 1) * n // parts
        wslot, row0, col0 = -1, 0, 0
        touched = set()
        for c in order[a:b]:
            s4, x4, y4 = slot[c0 + c], ix[c0 + c], iy[c0 + c]
            keys = []
            em_lanes = 0
            for ln in range(16):
                pend = None
                lane_em = False
                for j in range(4):
                    s = s4[ln, j]
                    if s < 0:
                        continue
                    key = (int(s), int(y4[ln, j]), int(x4[ln, j]) >> 6)
                    if pend is not None and key != pend:
                        keys.append(pend); lane_em = True
                    pend = key
                if pend is not None:
                    keys.append(pend); lane_em = True
                em_lanes += lane_em
            miss_keys = []
            for key in keys:
                emitted += 1
                s, y, x = key
                if s == wslot and 0 <= y - row0 < win_rows and 0 <= x - col0 < win_cols:
                    touched.add((y, x))
                else:
                    misses += 1; miss_keys.append(key)
            if miss_keys and 2 * len(miss_keys) > em_lanes:
                if wslot >= 0:
                    flushes += 1; flushed += len(touched)
                touched = set()
                lowest = min(miss_keys)
                wslot = lowest[0]
                row0 = max(0, lowest[1] - 2)
                col0 = max(0, min(min(k[2] for k in miss_keys if k[0] == wslot), W64 - win_cols))
        if wslot >= 0:
            flushes += 1; flushed += len(touched)
    return emitted, misses, flushes, flushed


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    oracle = ob.load_oracle()
    sc_list = scenes.batch_scenes(ssd, W, H, n, base_seed=100000, rng_seed=1000)
    trans = ssd.transformation_for_scene(sc_list[0])
    cfg = ssd.default_config(W, H)
    ocfg, ocal = ob.to_oracle_config(cfg), ob.to_oracle_calibration(trans.constants)
    A, b = np.array(trans.constants.a).reshape(3, 3), np.array(trans.constants.b)
    variants = [(64, 4, "half"), (64, 4, "pre"), (128, 4, "pre"), (32, 8, "pre"), (128, 2, "pre")]
    tot = {v: np.zeros(4) for v in variants}
    for sc in sc_list:
        slot, ix, iy, want = frame_points(oracle, ocfg, ocal, A, b, sc)
        for v in variants:
            tot[v] += simulate(slot, ix, iy, want, *v)
    for v in variants:
        e, m, f, fw = tot[v] / n
        print("window %3d rows x %d words, rule %-8s: %6.0f words emitted, %6.0f miss (%4.1f %%), %4.0f flushes, %5.0f words flushed -> %6.0f global atomics" %
              (v[0], v[1], v[2], e, m, 100 * m / e, f, fw, m + fw))


if __name__ == "__main__":
    main()
