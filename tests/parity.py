"""Parity checker: HIP path (through the C ABI) vs the CPU oracle on the same frame.

TEST INFRASTRUCTURE (uses oracle/): imported by tests/, __graft_entry__.smoke() and bench.py's
parity spot-checks only.

Bars:
  * integers, indices, bytes (counts, histogram, peaks, plateau table, raw and closed images, scans,
    integer lines, probe points, serialized text): bit-exact.
  * doubles that come out of identical IEEE operation sequences on integer inputs: bounds and base line compared to
    1e-12 absolute; the chosen vertical-edge points, vertical lines and all corners (image, world, external world):
    IDENTICAL (tolerance 0) — including the order std::sort leaves among equal distances.
  * mean z / step height: the device accumulates round(z*2^40) in int64 (order-independent, bitwise
    reproducible) where the reference adds doubles in point order: |diff| <= 1e-9 m
    (north-star bar: 1e-4 m).
"""
import numpy as np

import oracle_binding as ob

TOL_GEOM = 1e-12
TOL_EXACT = 0.0        # corners, lines: identical doubles
TOL_HEIGHT = 1e-9


class Mismatch(AssertionError):
    pass


def _eq(name, a, b):
    if a != b:
        raise Mismatch("%s: device %r != oracle %r" % (name, a, b))


def _close(name, a, b, tol):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    both_nan = np.isnan(a) & np.isnan(b)
    d = np.where(both_nan, 0.0, np.abs(a - b))
    if not np.all(d <= tol):
        raise Mismatch("%s: max |diff| %.3e > %.1e (device %r, oracle %r)" % (name, float(np.nanmax(d)), tol, a.tolist(), b.tolist()))
    return float(np.max(d)) if d.size else 0.0


def compare_debug(dbg, res, report):
    """ssd.DebugFrame vs oracle Result: every intermediate."""
    _eq("n_nonzero", dbg.n_nonzero, res.n_nonzero)
    _eq("n_inrange", dbg.n_inrange, res.n_inrange)
    _eq("n_bins", dbg.n_bins, res.n_bins)
    _eq("min_height", dbg.min_height, res.min_height)
    _eq("min_img_y_extent", dbg.min_img_y_extent, res.min_img_y_extent)
    _eq("hist", list(dbg.hist[:res.n_bins]), list(res.hist[:res.n_bins]))
    _eq("n_peaks", dbg.n_peaks, res.n_peaks)
    _eq("peaks", list(dbg.peaks[:res.n_peaks]), list(res.peaks[:res.n_peaks]))
    _eq("n_plateaus", dbg.n_plateaus, res.n_plateaus)
    _eq("ground_ind", dbg.ground_ind, res.ground_ind)
    _eq("first_valid_ind", dbg.first_valid_ind, res.first_valid_ind)
    _eq("n_oob", dbg.n_oob, res.n_oob)
    worst = 0.0
    for k in range(res.n_plateaus):
        d, o = dbg.plateaus[k], res.plateaus[k]
        tag = "plateau[%d]" % k
        _eq(tag + ".peak_bin", d.peak_bin, o.peak_bin)
        _eq(tag + ".pair", (d.bin_lo, d.bin_hi), (o.bin_lo, o.bin_hi))
        _eq(tag + ".n_points", d.n_points, o.n_points)
        _eq(tag + ".is_step", d.is_step, o.is_step)
        if not o.is_step:
            continue
        _eq(tag + ".n_scans", (d.n_scans_right, d.n_scans_left), (o.n_scans_right, o.n_scans_left))
        _eq(tag + ".scans_right", [list(r) for r in d.scans_right[:o.n_scans_right]], [list(r) for r in o.scans_right[:o.n_scans_right]])
        _eq(tag + ".scans_left", [list(r) for r in d.scans_left[:o.n_scans_left]], [list(r) for r in o.scans_left[:o.n_scans_left]])
        _eq(tag + ".outline_found", d.outline_found, o.outline_found)
        _eq(tag + ".valid", d.valid, o.valid)
        if o.outline_found:
            _eq(tag + ".n_edge_pts", list(d.n_edge_pts), list(o.n_edge_pts))
            _eq(tag + ".lines", [list(l) for l in d.line], [list(l) for l in o.line])
            worst = max(worst, _close(tag + ".bounds", np.array(d.bounds), np.array(o.bounds), TOL_GEOM))
            worst = max(worst, _close(tag + ".base_line", list(d.base_line), list(o.base_line), TOL_GEOM))
            _eq(tag + ".vedge_found", list(d.vedge_found), list(o.vedge_found))
            _eq(tag + ".n_vpts", list(d.n_vpts), list(o.n_vpts))
            for s in range(2):
                _eq(tag + ".vpts[%d]" % s, [list(p) for p in d.vpts[s][:o.n_vpts[s]]], [list(p) for p in o.vpts[s][:o.n_vpts[s]]])
                if o.vedge_found[s]:
                    # std::sort is not stable: the kernel reproduces libstdc++'s order where equal distances give different lines
                    # (the chosen point itself may differ when they do not)
                    worst = max(worst, _close(tag + ".vline[%d]" % s, list(d.vline[s]), list(o.vline[s]), TOL_EXACT))
            _eq(tag + ".corner_found", list(d.corner_found), list(o.corner_found))
        worst = max(worst, _close(tag + ".quad_img", list(d.quad_img), list(o.quad_img), TOL_EXACT))
        worst = max(worst, _close(tag + ".quad_world", list(d.quad_world), list(o.quad_world), TOL_EXACT))
        if o.valid and res.first_valid_ind >= 0 and not (res.status & ob.ST_THROW):
            _eq(tag + ".n_in_quad", d.n_in_quad, o.n_in_quad)
            report["max_height_err"] = max(report.get("max_height_err", 0.0), _close(tag + ".mean_z", d.mean_z, o.mean_z, TOL_HEIGHT))
    if res.first_valid_ind >= 0 and res.ground_ind >= 0 and not (res.status & ob.ST_THROW):
        worst = max(worst, _close("ground_quad_world", list(dbg.ground_quad_world), list(res.ground_quad_world), TOL_EXACT))
        _eq("ground_n_in_quad", dbg.ground_n_in_quad, res.ground_n_in_quad)
        _eq("ground_front_valid", dbg.ground_front_valid, res.ground_front_valid)
        _eq("ground_n_pts", dbg.ground_n_pts, res.ground_n_pts)
        _eq("ground_pts", [list(p) for p in dbg.ground_pts[:res.ground_n_pts]], [list(p) for p in res.ground_pts[:res.ground_n_pts]])
        if res.ground_front_valid:
            _eq("ground_line", list(dbg.ground_line), list(res.ground_line))
            worst = max(worst, _close("ground_front_img", list(dbg.ground_front_img), list(res.ground_front_img), TOL_GEOM))
            report["max_height_err"] = max(report.get("max_height_err", 0.0), _close("ground_mean_z", dbg.ground_mean_z, res.ground_mean_z, TOL_HEIGHT))
    report["max_geom_err"] = max(report.get("max_geom_err", 0.0), worst)


def compare_result(ssd, fr, res, report):
    """ssd.FrameResult vs oracle Result: steps, status and the serialized line."""
    _eq("status", fr.status, res.status)
    _eq("n_steps", fr.n_steps, res.n_steps)
    for i in range(res.n_steps):
        o = res.steps_ext[i]
        report["max_height_err"] = max(report.get("max_height_err", 0.0),
                                       _close("step[%d].height" % i, fr.steps[i].height, o[0], TOL_HEIGHT))
        report["max_corner_err"] = max(report.get("max_corner_err", 0.0),
                                       _close("step[%d].quad" % i, list(fr.steps[i].quad), list(o[1:9]), TOL_EXACT))
    line = ssd.Stairs(fr).serialize()
    _eq("line", line, res.line.decode())
    report["line"] = line


def check_frame(ssd, oracle, det, cfg, cal, xyz, images=True, report=None, depth_intr=None):
    """Runs one frame through the HIP path (debug capture on) and the oracle and compares everything.
    With depth_intr, `xyz` is a uint16 depth frame: the HIP path deprojects on the fly (ssd_process_depth_host),
    the oracle deprojects first.  Returns the report dict; raises Mismatch on the first difference."""
    report = {} if report is None else report
    ocfg, ocal = ob.to_oracle_config(cfg), ob.to_oracle_calibration(cal)
    if depth_intr is not None:
        det.set_intrinsics(depth_intr)
    run = (lambda: det.process_depth_host(xyz)[0]) if depth_intr is not None else (lambda: det.process_host(xyz)[0])
    # records only: the kernels exactly as in production (k_inquad rasters only the ground pixels the bottom scan reads)
    det.set_debug(True, images=False)
    fr = run()
    dbg = det.debug(0)
    n_img = ssd.MAX_STEP_IMAGES if images else 0
    oxyz = oracle.deproject(depth_intr, xyz) if depth_intr is not None else xyz
    res, raw, closed, graw, gclosed = oracle.process(ocfg, ocal, oxyz, images=n_img, ground_images=images)
    compare_debug(dbg, res, report)
    compare_result(ssd, fr, res, report)
    if images:
        # again with image capture (the whole ground image rastered): the same records and results, and every image
        det.set_debug(True, images=True)
        fr2 = run()
        compare_debug(det.debug(0), res, report)
        compare_result(ssd, fr2, res, report)
        if bytes(fr2) != bytes(fr):
            raise Mismatch("the result differs between debug capture with and without images")
        n_step_imgs = sum(1 for k in range(res.n_plateaus) if res.plateaus[k].is_step)
        for s in range(min(n_step_imgs, ssd.MAX_STEP_IMAGES)):
            if not np.array_equal(det.debug_image(0, s, False), raw[s]):
                raise Mismatch("raw image of step plateau %d differs (%d pixels)" % (s, int((det.debug_image(0, s, False) != raw[s]).sum())))
            if not np.array_equal(det.debug_image(0, s, True), closed[s]):
                raise Mismatch("closed image of step plateau %d differs (%d pixels)" % (s, int((det.debug_image(0, s, True) != closed[s]).sum())))
        if res.first_valid_ind >= 0 and res.ground_ind >= 0 and not (res.status & ob.ST_THROW):
            if not np.array_equal(det.debug_image(0, -1, False), graw):
                raise Mismatch("raw ground image differs")
            if not np.array_equal(det.debug_image(0, -1, True), gclosed):
                raise Mismatch("closed ground image differs")
        report["images_checked"] = n_step_imgs + 1
    det.set_debug(False)
    report["n_steps"] = res.n_steps
    report["status"] = res.status
    return report


def check_results_only(ssd, oracle, cfg, cal, xyz, fr, report=None):
    """Compares a FrameResult obtained elsewhere (batch path) with the oracle's lean run."""
    ocfg, ocal = ob.to_oracle_config(cfg), ob.to_oracle_calibration(cal)
    return compare_results_only(ssd, oracle, fr, oracle.process_lean(ocfg, ocal, xyz), report)


def compare_results_only(ssd, oracle, fr, lean, report=None):
    """a FrameResult against what oracle.process_lean returned for the same frame: (n, steps, status)"""
    report = {} if report is None else report
    n, steps, status = lean
    _eq("status", fr.status, status)
    _eq("n_steps", fr.n_steps, n)
    for i in range(n):                       # corners first: a moved corner also moves the height, not the other way round
        report["max_corner_err"] = max(report.get("max_corner_err", 0.0),
                                       _close("step[%d].quad" % i, list(fr.steps[i].quad), list(steps[i][1:9]), TOL_EXACT))
    for i in range(n):
        report["max_height_err"] = max(report.get("max_height_err", 0.0),
                                       _close("step[%d].height" % i, fr.steps[i].height, steps[i][0], TOL_HEIGHT))
    if not (status & ob.ST_THROW):
        _eq("line", ssd.Stairs(fr).serialize(), oracle.serialize(steps) if n else '["stairs",["stairSteps",0]]')
    return report


def check_batch_against_oracle(ssd, oracle, cfg, cal, buf, frame_bytes, results, width, height, workers=None, chunk=64, report=None):
    """EVERY frame of a batch resident in device memory (frame i at buf.ptr + i * frame_bytes, float xyz) against the oracle:
    the frames come back from the device `chunk` at a time, the oracle runs on a pool of threads (re-entrant; ctypes drops the
    GIL: 11 ms per XGA frame and core), the comparisons in the calling thread.  Returns the number of frames checked."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    report = {} if report is None else report
    ocfg, ocal = ob.to_oracle_config(cfg), ob.to_oracle_calibration(cal)
    workers = workers or max(1, min(16, len(os.sched_getaffinity(0))))
    checked = 0
    with ThreadPoolExecutor(workers) as pool:
        for at in range(0, len(results), chunk):
            n = min(chunk, len(results) - at)
            host = buf.download(frame_bytes * n, offset=frame_bytes * at, dtype=np.float32).reshape(n, height * width, 3)
            for k, lean in enumerate(pool.map(lambda x: oracle.process_lean(ocfg, ocal, x), [host[k] for k in range(n)])):
                try:
                    compare_results_only(ssd, oracle, results[at + k], lean, report)
                except Mismatch as e:
                    raise Mismatch("frame %d: %s" % (at + k, e))
                checked += 1
    report["frames_checked"] = checked
    return checked


def compare_risers(dev, ora, report=None):
    """ssd.FrameRisers vs the oracle's list of Riser (extension: vertical faces).  Counts exact; geometry is the steps'
    (identical doubles expected); mean offset: fixed-point sum on both sides."""
    report = {} if report is None else report
    _eq("n_risers", dev.n_risers, len(ora))
    for i, o in enumerate(ora):
        d = dev.risers[i]
        tag = "riser[%d]" % i
        _eq(tag + ".n_points", d.n_points, o.n_points)
        _eq(tag + ".detected", d.detected, o.detected)
        _close(tag + ".heights", [d.height_bottom, d.height_top], [o.height_bottom, o.height_top], TOL_HEIGHT)
        _close(tag + ".edge", list(d.left) + list(d.right), list(o.left) + list(o.right), TOL_EXACT)
        report["max_offset_err"] = max(report.get("max_offset_err", 0.0), _close(tag + ".mean_offset", d.mean_offset, o.mean_offset, 1e-12))
    report["risers_detected"] = sum(1 for o in ora if o.detected)
    return report
