"""Generates tests/golden/ref_serialize.json, ref_quadtest.json, ref_configuration.json, ref_lines.json and ref_print_stairs.json by running
the REAL reference code on seeded inputs: oracle/_ref/libssd_ref.so (= /root/reference/stairs.cpp + quadrilateralTest.cpp +
configuration.h compiled in place by oracle/Makefile), and /root/reference/print-stairs.py — the reference's own consumer of
the stdout line — run as a subprocess where it lies, fed the lines of tests/golden/oracle_goldens.json on stdin.
Run in the build container (needs /root/reference):

    make -C oracle && python tests/golden/make_ref_goldens.py

The fixtures are data only (inputs + the reference's outputs); doubles are stored as hex strings so
that they round-trip bit-exactly.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_binding as ob  # noqa: E402


def hexlist(a):
    # float.hex() drops a NaN's sign; float.fromhex("-nan") restores it (the sign is what "-nan" vs "nan" in a line depends on)
    return [("-nan" if np.signbit(x) else "nan") if np.isnan(x) else float(x).hex() for x in np.asarray(a, dtype=np.float64).reshape(-1)]


def random_quad(rng, kind):
    if kind == 0:      # near-rectangular tread, small yaw, metres
        cx, cy = rng.uniform(-0.2, 0.2), rng.uniform(0.3, 1.1)
        w, d, a = rng.uniform(0.2, 0.5), rng.uniform(0.08, 0.2), rng.uniform(-0.3, 0.3)
        c, s = np.cos(a), np.sin(a)
        base = np.array([[-w, -d], [w, -d], [-w, d], [w, d]]) + rng.normal(0, 0.01, (4, 2))
        return base @ np.array([[c, s], [-s, c]]) + [cx, cy]
    if kind == 1:      # arbitrary quads (many non-convex -> the constructor throws)
        return rng.uniform(-1, 1, (4, 2))
    if kind == 2:      # axis-aligned rectangles (degenerate cell maps)
        x0, x1 = sorted(rng.uniform(-1, 1, 2))
        y0, y1 = sorted(rng.uniform(-1, 1, 2))
        return np.array([[x0, y0], [x1, y0], [x0, y1], [x1, y1]])
    # strongly rotated convex quads (diamonds, slivers)
    a = rng.uniform(0, np.pi)
    c, s = np.cos(a), np.sin(a)
    w, d = rng.uniform(0.05, 0.8), rng.uniform(0.05, 0.8)
    base = np.array([[-w, -d], [w, -d], [-w, d], [w, d]]) * (1 + rng.normal(0, 0.05, (4, 2)))
    return base @ np.array([[c, s], [-s, c]])


def main():
    ref = ob.load_ref()
    if ref is None:
        raise SystemExit("oracle/_ref/libssd_ref.so missing: run `make -C oracle` where /root/reference exists")
    rng = np.random.default_rng(20211)

    ser = []
    specials = [0.0, -0.0, 0.0005, -0.0005, 0.0015, 0.0025, -0.0004999, 1e-9, -1e-9, 123456.789, -99999.9995,
                float("nan"), float("inf"), -float("inf"), 0.17005879162516252, -0.40078125000000003]
    for n in (0, 1, 2, 4, 9, 17):
        steps = rng.normal(0, 1, (n, 9))
        ser.append(steps)
    sp = np.array(specials + specials[:2])[:18].reshape(2, 9)
    ser.append(sp)
    ser.append(rng.choice(specials, (3, 9)))
    ser.append(np.round(rng.normal(0, 2, (5, 9)), 3) + 0.0005)     # ties of the 3-decimal rounding
    # round 4: the x86 default NaN (sign bit set) — what calcAverageZ's 0.0 / 0 leaves in a height — prints "-nan"; drawn from no rng
    neg_nan = np.frombuffer(np.array([0xfff8000000000000], np.uint64).tobytes(), np.float64)[0]
    ser.append(np.array([[neg_nan, -0.4, 0.45, 0.4, 0.45, -0.4, 0.73, 0.4, 0.73], [0.17, neg_nan, float("nan"), -neg_nan, 0.0, 1.0, 2.0, 3.0, 4.0]]))
    out_ser = [{"n": int(len(s)), "steps": hexlist(s), "line": ref.serialize(s)} for s in ser]
    with open(os.path.join(HERE, "ref_serialize.json"), "w") as f:
        json.dump(out_ser, f, indent=0)

    out_q = []
    degenerate = [
        [[0, 0], [1, 0], [0, 0], [1, 0]],             # no extent along Y
        [[0, 0], [0, 0], [0, 1], [0, 1]],             # no extent along X
        [[0, 0], [1, 0], [0, 1], [0, 1]],             # duplicate vertex (triangle)
        [[0, 0], [1, 1], [2, 2], [3, 3]],             # collinear
        [[0, 0], [1, 0], [1, 1], [0, 1]],             # bow-tie in the default vertex order
        [[0, 0], [0, 0], [0, 0], [0, 0]],             # a point
        [[-0.6, 1.3], [-0.6, 1.3], [-0.6, 1.3], [-0.6, 1.3]],
    ]
    for i in range(72 + len(degenerate)):
        q = np.array(degenerate[i - 72], dtype=np.float64) if i >= 72 else random_quad(rng, i % 4)
        lo, hi = q.min(0), q.max(0)
        ext = (hi - lo) * 0.25 + 1e-3
        pts = rng.uniform(lo - ext, hi + ext, (40, 2))
        # points on and next to vertices / edges / bbox sides
        t = rng.uniform(0, 1, (8, 1))
        e = [(0, 1), (1, 3), (3, 2), (2, 0)]
        on_edges = np.concatenate([q[a] + t[2 * k:2 * k + 2] * (q[b] - q[a]) for k, (a, b) in enumerate(e)])
        near = on_edges + rng.normal(0, 1e-12, on_edges.shape)
        pts = np.concatenate([pts, q, on_edges, near, np.nextafter(q, np.inf), np.nextafter(q, -np.inf)])
        rc, inside = ref.quad_test(q, pts)
        out_q.append({"quad": hexlist(q), "pts": hexlist(pts), "rc": int(rc), "inside": "".join(str(int(v)) for v in inside) if rc == 0 else ""})
    with open(os.path.join(HERE, "ref_quadtest.json"), "w") as f:
        json.dump(out_q, f, indent=0)
    print("serialize cases:", len(out_ser), "quad cases:", len(out_q), "throwing:", sum(1 for c in out_q if c["rc"] != 0),
          {c["rc"] for c in out_q})

    # configuration.h:27-52, default-constructed by the reference's own compiler-generated constructor
    vals, wh = ref.configuration()
    names = ["x_min", "x_max", "y_min", "y_max", "z_min", "z_max", "height_interval", "min_height_above_ground", "min_step_depth"]
    with open(os.path.join(HERE, "ref_configuration.json"), "w") as f:
        json.dump({"what": "stairs::Configuration{} of /root/reference/configuration.h:27-52 (doubles as hex)",
                   "values": {n: float(v).hex() for n, v in zip(names, vals)}, "depth_stream": {"width": wh[0], "height": wh[1]}}, f, indent=1)

    # types.h:117-163 — LineCoordinates<T>: the line through two points (int and double) and det / detx / dety of two lines
    cases = []
    specials = [0.0, -0.0, 1.0, -1.0, 0.1, 1e-300, 1e300, float("inf"), float("nan"), 0.17005879162516252, 1023.0, 767.0]
    for k in range(400):
        if k < 300:
            pq = rng.normal(0, [1, 1, 1, 1][k % 4] * 10.0 ** int(rng.integers(-3, 4)), 4)
        else:
            pq = rng.choice(specials, 4)
        cases.append({"pq": hexlist(pq), "abc": hexlist(ref.line(pq))})
    icases = []
    for k in range(300):
        pq = rng.integers(-2 ** (4 + k % 12), 2 ** (4 + k % 12), 4).astype(np.int32)       # up to 2^15: products stay inside int32 as in the path
        icases.append({"pq": [int(v) for v in pq], "abc": [int(v) for v in ref.line(pq, integer=True)]})
    dcases = []
    for k in range(400):
        l, o = (rng.normal(0, 10.0 ** int(rng.integers(-2, 5)), 3), rng.normal(0, 10.0 ** int(rng.integers(-2, 5)), 3)) if k < 320 else (rng.choice(specials, 3), rng.choice(specials, 3))
        dcases.append({"l": hexlist(l), "o": hexlist(o), "dets": hexlist(ref.line_dets(l, o))})
    with open(os.path.join(HERE, "ref_lines.json"), "w") as f:
        json.dump({"what": "stairs::LineCoordinates<T> of /root/reference/types.h:117-163 (doubles as hex): coefficients of the line through p, q; det / detx / dety of two lines",
                   "double": cases, "int": icases, "dets": dcases}, f, indent=0)
    print("lines:", len(cases), len(icases), len(dcases))

    # print-stairs.py:53-77 — the reference's terminal consumer of the line: what IT reads out of each golden line
    import subprocess
    gold = json.load(open(os.path.join(HERE, "oracle_goldens.json")))["frames"]
    lines = [(name, fr["line"]) for name, fr in sorted(gold.items()) if fr.get("line")]
    proc = subprocess.run([sys.executable, "/root/reference/print-stairs.py"], input="".join(l + "\n" for _, l in lines),
                          capture_output=True, text=True, check=True)
    with open(os.path.join(HERE, "ref_print_stairs.json"), "w") as f:
        json.dump({"what": "stdout of /root/reference/print-stairs.py (run unmodified, where it lies) fed the golden lines below, one per line, on stdin",
                   "frames": [n for n, _ in lines], "stdin_lines": [l for _, l in lines], "stdout": proc.stdout}, f, indent=1)
    print("configuration:", dict(zip(names, vals)), wh, "| print-stairs.py: %d lines in, %d bytes out" % (len(lines), len(proc.stdout)))


if __name__ == "__main__":
    main()
