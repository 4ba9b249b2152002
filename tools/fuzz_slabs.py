#!/usr/bin/env python3
"""tools/fuzz_slabs.py [N_FRAMES] [SEED] — randomised parity sweep on unstructured input: each frame is a handful of
random horizontal slabs (rotated rectangles at random heights, point counts around the 2000-point threshold, thickness
jitter that spreads them over several 1 cm bins, some outside the measuring range) plus scattered points, at random
positions of the frame.  Exercises the histogram / peak / plateau-pair logic (quirks Q1-Q4) and degenerate outlines far
away from anything a staircase produces.  Identity calibration shifted by 0.5 m in z, as tests/test_gpu_quirks.py.
TEST INFRASTRUCTURE (uses oracle/)."""
import importlib
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
ssd = importlib.import_module("stair-step-detector_amd")
import oracle_binding as ob  # noqa: E402
import parity  # noqa: E402

W, H = 640, 480
Z_SHIFT = 0.5


def frame(rng):
    pts = []
    for _ in range(int(rng.integers(1, 13))):
        z = float(rng.uniform(-0.13, 1.13)) if rng.random() < 0.8 else float(rng.integers(-10, 110)) / 100.0 + float(rng.choice([0.0, 0.005, 1e-9]))
        n = int(rng.choice([rng.integers(200, 2500), rng.integers(1900, 2100), rng.integers(2500, 60000)]))
        cx, cy = float(rng.uniform(-0.5, 0.5)), float(rng.uniform(0.2, 1.2))
        hx, hy = float(rng.uniform(0.05, 0.6)), float(rng.uniform(0.03, 0.5))
        ang = float(rng.uniform(-0.8, 0.8)) if rng.random() < 0.5 else 0.0
        u, v = rng.uniform(-hx, hx, n), rng.uniform(-hy, hy, n)
        if rng.random() < 0.5:                                     # regular grid: rasterises to a solid block
            nx = max(1, int(np.sqrt(n * hx / hy)))
            gu, gv = np.meshgrid(np.linspace(-hx, hx, nx), np.linspace(-hy, hy, (n + nx - 1) // nx))
            u, v = gu.ravel()[:n], gv.ravel()[:n]
        x = cx + u * np.cos(ang) - v * np.sin(ang)
        y = cy + u * np.sin(ang) + v * np.cos(ang)
        thick = float(rng.choice([0.0, 0.0, 0.002, 0.008, 0.02]))
        zz = z + rng.normal(0.0, thick, len(x)) if thick > 0 else np.full(len(x), z)
        pts.append(np.stack([x, y, zz], 1))
    if rng.random() < 0.5:
        m = int(rng.integers(100, 20000))
        pts.append(np.stack([rng.uniform(-0.7, 0.7, m), rng.uniform(0.0, 1.4, m), rng.uniform(-0.2, 1.2, m)], 1))
    p = np.concatenate(pts)
    if len(p) > W * H:
        p = p[rng.permutation(len(p))[:W * H]]
    out = np.zeros((W * H, 3), dtype=np.float32)
    idx = np.sort(rng.permutation(W * H)[:len(p)])
    if rng.random() < 0.5:
        p = p[np.lexsort((p[:, 0], -p[:, 1]))]                    # camera-like order: far rows first, left to right
    p = p.copy()
    p[:, 2] += Z_SHIFT
    out[idx] = p.astype(np.float32)
    return out.reshape(H, W, 3)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    oracle = ob.load_oracle()
    trans = ssd.GeometricTransformation()
    trans.constants.b[2] = -Z_SHIFT
    B = 32
    cfg = ssd.default_config(W, H, max_frames_per_batch=B)
    det = ssd.Detector(cfg, trans, 0)
    total = bad = 0
    hist, failures = {}, []
    worst = {"max_height_err": 0.0, "max_corner_err": 0.0}
    flags = {"oob_pixel": 0, "assert": 0, "overflow": 0}
    for b0 in range(0, n, B):
        frames = np.stack([frame(rng) for _ in range(min(B, n - b0))])
        res = det.process_host(frames)
        res = [type(r).from_buffer_copy(bytes(r)) for r in res]

        def check(i):
            rep = {}
            try:
                parity.check_results_only(ssd, oracle, cfg, trans.constants, frames[i], res[i], rep)
                return i, None, rep
            except parity.Mismatch as e:
                return i, str(e), rep
        with ThreadPoolExecutor(min(len(os.sched_getaffinity(0)), 32)) as pool:
            for i, err, rep in pool.map(check, range(len(frames))):
                total += 1
                st = res[i].status
                key = "throw" if st & 1 else str(res[i].n_steps)
                hist[key] = hist.get(key, 0) + 1
                flags["oob_pixel"] += 1 if st & 2 else 0
                flags["assert"] += 1 if st & 4 else 0
                flags["overflow"] += 1 if st & 8 else 0
                for k in worst:
                    worst[k] = max(worst[k], rep.get(k, 0.0))
                if err:
                    bad += 1
                    failures.append({"frame": b0 + i, "error": err[:300]})
                    np.save(os.path.join(ROOT, "gpurun_out", "slab_fail_%d_%d.npy" % (seed, b0 + i)), frames[i])
    det.close()
    print(json.dumps({"frames": total, "mismatches": bad, "steps_histogram": hist, "status_flags": flags, **worst, "failures": failures[:10]}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
