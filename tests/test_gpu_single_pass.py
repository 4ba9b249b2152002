"""The single pass: K1 rasters the step plateaus itself, into planes of the height bins a sampled histogram (k_predict) expects them
in; k_peaks checks the planes against the plateaus of the complete histogram frame by frame, and k_raster does the frames they do
not cover.  Whatever the predictor says, the results are those of the two-pass pipeline, bit for bit, and the planes are left
zero."""
import numpy as np
import pytest

import parity
import scenes

pytestmark = pytest.mark.gpu


def _run(det, buf, n):
    det.enqueue(buf.ptr, n)
    return [bytes(x) for x in det.fetch_list(n)]


def _batch(ssd, gpu_device, W, H, n, base_seed, rng_seed):
    sc = scenes.batch_scenes(ssd, W, H, n, base_seed=base_seed, rng_seed=rng_seed)
    buf = ssd.DeviceBuffer(W * H * 12 * n, gpu_device)
    ssd.synth_device(sc, buf.ptr, device=gpu_device)
    return sc, buf


def test_single_pass_covers_the_batch_and_matches_the_two_pass_pipeline(ssd, oracle, gpu_device):
    """256 XGA staircase frames: the default handle runs the single pass on them (a batch of at least 64 XGA frames' worth of points), the predictor covers
    nearly every frame, the planes are zero afterwards, and the results equal those of the same handle with the single pass
    switched off - and the oracle's, every frame."""
    W, H, n = 1024, 768, 256
    sc, buf = _batch(ssd, gpu_device, W, H, n, 41000, 41)
    trans = ssd.transformation_for_scene(sc[0])
    cfg = ssd.default_config(W, H, max_frames_per_batch=n)
    det = ssd.Detector(cfg, trans, gpu_device)
    single = _run(det, buf, n)
    st = det.single_pass_stats(n)
    assert st["ran"] and st["dirty_words"] == 0
    assert st["with_steps"] >= n * 9 // 10
    assert st["covered"] >= st["with_steps"] * 9 // 10, st
    assert st["with_steps"] <= st["planes"] <= ssd.MAX_PLANES * n
    assert _run(det, buf, n) == single                       # the planes and their boxes were left clean
    det.single_pass(0)
    assert _run(det, buf, n) == single
    assert not det.single_pass_stats(n)["ran"]
    det.single_pass(-1)
    det.enqueue(buf.ptr, n)
    res = det.fetch_list(n)
    assert parity.check_batch_against_oracle(ssd, oracle, cfg, trans.constants, buf, W * H * 12, res, W, H) == n
    det.close()
    buf.free()


@pytest.mark.parametrize("sabotage", [1, 2])
def test_a_wrong_predictor_costs_time_not_results(ssd, gpu_device, sabotage):
    """k_predict sabotaged (its planes three bins above the right ones / no planes at all): K1 rasters the wrong bins or nothing,
    k_peaks finds the plateaus uncovered, k_raster does every frame, k_outline clears the stray planes."""
    W, H, n = 1024, 768, 128
    sc, buf = _batch(ssd, gpu_device, W, H, n, 42000, 42)
    det = ssd.Detector(ssd.default_config(W, H, max_frames_per_batch=n), ssd.transformation_for_scene(sc[0]), gpu_device)
    det.single_pass(0)
    two_pass = _run(det, buf, n)
    det.single_pass(-1, sabotage)
    assert _run(det, buf, n) == two_pass
    st = det.single_pass_stats(n)
    assert st["ran"] and st["dirty_words"] == 0 and st["with_steps"] >= n * 9 // 10
    assert st["covered"] == 0 if sabotage == 2 else st["covered"] < st["with_steps"] // 4, st
    assert (st["planes"] == 0) == (sabotage == 2)
    det.single_pass(-1, 0)
    assert _run(det, buf, n) == two_pass
    assert det.single_pass_stats(n)["covered"] >= st["with_steps"] * 9 // 10
    det.close()
    buf.free()


@pytest.mark.parametrize("W,H", [(1024, 768), (512, 384), (256, 192), (640, 480), (848, 480), (1280, 720), (1920, 1080), (600, 450), (427, 321), (128, 100)])
def test_single_pass_forced_on_small_batches(ssd, oracle, gpu_device, W, H):
    """Geometries whose tile of 1024 points is one, two or four camera rows (K1 walks its chunk tile by tile) and geometries where
    it is not (K1 walks the chunk's strips of 256 points sorted by column band; 848 is not even a whole number of cells; 600 x 450
    and 427 x 321 end in a part of a strip - the clamped loads of round 5 - and the latter is not even a multiple of four points:
    12-byte loads; 128 x 100 ends in half a tile), 12
    frames with the single pass forced on (the product only takes it for batches of 64 XGA frames' worth of points and more), against two passes and the oracle."""
    n = 12
    sc, buf = _batch(ssd, gpu_device, W, H, n, 43000 + W, 43)
    trans = ssd.transformation_for_scene(sc[0])
    cfg = ssd.default_config(W, H, max_frames_per_batch=n)
    det = ssd.Detector(cfg, trans, gpu_device)
    det.single_pass(0)
    two_pass = _run(det, buf, n)
    det.single_pass(1)
    assert _run(det, buf, n) == two_pass
    st = det.single_pass_stats(n)
    assert st["ran"] and st["dirty_words"] == 0
    det.enqueue(buf.ptr, n)
    res = det.fetch_list(n)
    assert parity.check_batch_against_oracle(ssd, oracle, cfg, trans.constants, buf, W * H * 12, res, W, H) == n
    if W in (1024, 640):
        # one frame with debug capture, still forced: every record, the raw and the closed image of every step plateau (read
        # from the planes by k_outline) and the ground image against the oracle's
        xyz = buf.download(W * H * 12, dtype=np.float32).reshape(H, W, 3)
        rep = parity.check_frame(ssd, oracle, det, cfg, trans.constants, xyz, images=True)
        assert rep["images_checked"] >= 2 and det.single_pass_stats(1)["ran"]
    det.close()
    buf.free()


def test_single_pass_on_a_vga_batch_with_depth_input(ssd, oracle, gpu_device):
    """640 x 480 (a tile is 1.6 camera rows: the sorted strips), 192 frames as the product runs them (it takes the single pass from
    64 XGA frames' worth of points on: 164 VGA frames), vertices and 16-bit depth"""
    W, H, n = 640, 480, 192
    sc, buf = _batch(ssd, gpu_device, W, H, n, 44000, 44)
    trans = ssd.transformation_for_scene(sc[0])
    cfg = ssd.default_config(W, H, max_frames_per_batch=n)
    det = ssd.Detector(cfg, trans, gpu_device)
    single = _run(det, buf, n)
    st = det.single_pass_stats(n)
    assert st["ran"] and st["dirty_words"] == 0 and st["covered"] >= st["with_steps"] * 8 // 10, st
    det.single_pass(0)
    assert _run(det, buf, n) == single
    det.single_pass(-1)
    det.enqueue(buf.ptr, n)
    res = det.fetch_list(n)
    assert parity.check_batch_against_oracle(ssd, oracle, cfg, trans.constants, buf, W * H * 12, res, W, H) == n
    intr = ssd.intrinsics_for_scene(sc[0])
    det.set_intrinsics(intr)
    dbuf = ssd.DeviceBuffer(W * H * 2 * n, gpu_device)
    ssd.synth_depth_device(sc, dbuf.ptr, device=gpu_device)
    det.enqueue_depth(dbuf.ptr, n)
    assert not det.single_pass_stats(n)["ran"]          # the product keeps two passes on depth input
    det.single_pass(1)
    det.enqueue_depth(dbuf.ptr, n)
    d1 = [bytes(x) for x in det.fetch_list(n)]
    assert det.single_pass_stats(n)["ran"]
    det.single_pass(0)
    det.enqueue_depth(dbuf.ptr, n)
    assert [bytes(x) for x in det.fetch_list(n)] == d1
    dbuf.free()
    det.close()
    buf.free()


def test_single_pass_on_depth_input_and_unaligned_vertices(ssd, oracle, gpu_device):
    """The other two sources of K1: 16-bit depth frames deprojected on the fly, and vertex frames whose stride is not a multiple
    of 16 bytes (12-byte loads) - 96 XGA frames each, single pass against two passes, depth also against the oracle."""
    W, H, n = 1024, 768, 96
    sc = scenes.batch_scenes(ssd, W, H, n, base_seed=45000, rng_seed=45)
    trans = ssd.transformation_for_scene(sc[0])
    cfg = ssd.default_config(W, H, max_frames_per_batch=n)
    det = ssd.Detector(cfg, trans, gpu_device)
    intr = ssd.intrinsics_for_scene(sc[0])
    det.set_intrinsics(intr)
    dbuf = ssd.DeviceBuffer(W * H * 2 * n, gpu_device)
    ssd.synth_depth_device(sc, dbuf.ptr, device=gpu_device)

    def run_depth():
        det.enqueue_depth(dbuf.ptr, n)
        return det.fetch_list(n)

    det.single_pass(0)
    two_pass = [bytes(x) for x in run_depth()]
    det.single_pass(1)                       # forced: the product keeps two passes on depth input (no gain there)
    res = run_depth()
    assert [bytes(x) for x in res] == two_pass
    st = det.single_pass_stats(n)
    assert st["ran"] and st["dirty_words"] == 0 and st["covered"] >= st["with_steps"] * 9 // 10
    depth = dbuf.download(W * H * 2 * n, dtype=np.uint16).reshape(n, H, W)
    for i in range(0, n, 8):
        parity.check_results_only(ssd, oracle, cfg, trans.constants, oracle.deproject(intr, depth[i]), res[i])
    dbuf.free()
    det.single_pass(-1)
    # vertices at a stride of 12 * W * H + 4 bytes: every second frame starts off a 16-byte boundary
    stride = W * H * 12 + 4
    xyz = ssd.synth_host(sc[:64])
    raw = np.zeros(64 * stride, dtype=np.uint8)
    for i in range(64):
        raw[i * stride:i * stride + W * H * 12] = xyz[i].view(np.uint8).reshape(-1)
    vbuf = ssd.DeviceBuffer(raw.size, gpu_device)
    vbuf.upload(raw)
    det.single_pass(0)
    det.enqueue(vbuf.ptr, 64, stride_bytes=stride)
    two_pass = [bytes(x) for x in det.fetch_list(64)]
    det.single_pass(-1)
    det.enqueue(vbuf.ptr, 64, stride_bytes=stride)
    assert [bytes(x) for x in det.fetch_list(64)] == two_pass
    st = det.single_pass_stats(64)
    assert st["ran"] and st["dirty_words"] == 0
    vbuf.free()
    det.close()


def test_the_predictor_is_timed_beside_the_stages(ssd, gpu_device):
    W, H, n = 1024, 768, 64
    sc, buf = _batch(ssd, gpu_device, W, H, n, 46000, 46)
    det = ssd.Detector(ssd.default_config(W, H, max_frames_per_batch=n), ssd.transformation_for_scene(sc[0]), gpu_device)
    det.set_timing(True)
    _run(det, buf, n)
    assert det.predict_time_ms() > 0.0 and det.stage_times_ms()["hist"] > 0.0
    det.single_pass(0)
    _run(det, buf, n)
    assert det.predict_time_ms() == 0.0 and det.predict_time_ms(back=1) > 0.0
    det.close()
    buf.free()


def test_a_batch_the_predictor_does_not_cover_switches_the_next_ones_to_two_passes(ssd, gpu_device):
    """ssd_fetch learns with the results how many frames k_raster had to do; more than half of a batch, and the following batches
    run two passes (63 of them, then one probes again) - a wrong predictor is paid for once, not per batch."""
    W, H, n = 1024, 768, 64
    sc, buf = _batch(ssd, gpu_device, W, H, n, 47000, 47)
    det = ssd.Detector(ssd.default_config(W, H, max_frames_per_batch=n), ssd.transformation_for_scene(sc[0]), gpu_device)
    good = _run(det, buf, n)
    assert det.single_pass_stats(n)["ran"]
    assert _run(det, buf, n) == good and det.single_pass_stats(n)["ran"]         # covered: stays on
    det.single_pass(-1, 2)                                                      # from now on the predictor gives no planes
    assert _run(det, buf, n) == good and det.single_pass_stats(n)["ran"]         # this batch pays for it
    for _ in range(3):
        assert _run(det, buf, n) == good and not det.single_pass_stats(n)["ran"]
    det.single_pass(-1, 0)                                                      # (the hook also ends the back-off)
    assert _run(det, buf, n) == good and det.single_pass_stats(n)["ran"]
    det.close()
    buf.free()


@pytest.mark.parametrize("W,H,n,sabotage", [(1024, 768, 128, 0), (1024, 768, 64, 1), (640, 480, 32, 0), (1920, 1080, 16, 0)])
def test_the_kernels_table_is_the_host_statements(ssd, gpu_device, W, H, n, sabotage):
    """k_predict makes its bin -> plane table with one thread per bin (ballots, rank counting); csrc/ssd_predict.h states it as a
    plain function (tests/test_predict.py checks that one on the CPU).  On every frame: the table the kernel left equals the
    function of the sample the kernel counted."""
    if W == 1920:
        sc = scenes.fhd_stress_scenes(ssd, n, base_seed=9100)
    else:
        sc = scenes.batch_scenes(ssd, W, H, n, base_seed=48000 + W, rng_seed=48)
    buf = ssd.DeviceBuffer(W * H * 12 * n, gpu_device)
    ssd.synth_device(sc, buf.ptr, device=gpu_device)
    cfg = ssd.default_config(W, H, max_frames_per_batch=n)
    det = ssd.Detector(cfg, ssd.transformation_for_scene(sc[0]), gpu_device)
    det.set_debug(True, images=False)                                        # the handle's bin count and lowest step bin, from a debug record
    det.enqueue(buf.ptr, 1)
    det.fetch(1)
    n_bins, min_height = det.debug(0).n_bins, det.debug(0).min_height
    det.set_debug(False)
    det.single_pass(1, sabotage)
    det.enqueue(buf.ptr, n)
    det.fetch(n)
    total_planes = 0
    for i in range(n):
        table, planes, _, _ = det.single_pass_frame(i)
        sample = det.single_pass_sample(i)
        assert int(sample.sum()) > W * H // 16 // 4                          # a sixteenth of the frame, most of it in range
        host, host_planes = ssd.predict_table_host(sample, n_bins, min_height, sabotage)
        assert planes == host_planes and np.array_equal(table, host), "frame %d" % i
        total_planes += planes
    assert total_planes >= n
    det.close()
    buf.free()


def test_batches_without_stairs_switch_to_two_passes(ssd, gpu_device):
    """No step plateau in sight: nothing for the single pass to gain (k_raster has nothing to do either way) - after one such batch
    the handle runs two passes; results equal either way."""
    W, H, n = 1024, 768, 64
    sc = [scenes.make(ssd, "xga_no_stairs")] * n
    buf = ssd.DeviceBuffer(W * H * 12 * n, gpu_device)
    ssd.synth_device(sc, buf.ptr, device=gpu_device)
    det = ssd.Detector(ssd.default_config(W, H, max_frames_per_batch=n), ssd.transformation_for_scene(sc[0]), gpu_device)
    first = _run(det, buf, n)
    st = det.single_pass_stats(n)
    assert st["ran"] and st["with_steps"] == 0 and st["dirty_words"] == 0
    for _ in range(2):
        assert _run(det, buf, n) == first and not det.single_pass_stats(n)["ran"]
    det.close()
    buf.free()


def test_the_planes_memory_can_be_given_back(ssd, gpu_device):
    """ssd_set_single_pass(0): 2.4 MB per XGA frame and workspace back, two passes from then on; (1): the default again"""
    W, H, n = 1024, 768, 64
    sc, buf = _batch(ssd, gpu_device, W, H, n, 49000, 49)
    det = ssd.Detector(ssd.default_config(W, H, max_frames_per_batch=n), ssd.transformation_for_scene(sc[0]), gpu_device)
    with_planes = det.workspace_bytes
    single = _run(det, buf, n)
    assert det.single_pass_stats(n)["ran"]
    det.set_single_pass(False)
    plane_bytes = ssd.plane_pool_size(n, W * H) * H * (W // 64) * 8
    assert det.workspace_bytes == with_planes - plane_bytes
    assert _run(det, buf, n) == single and not det.single_pass_stats(n)["ran"]
    det.set_single_pass(True)
    assert det.workspace_bytes == with_planes
    assert _run(det, buf, n) == single and det.single_pass_stats(n)["ran"]
    det.close()
    buf.free()
    # a handle too small for it: nothing to give or take
    small = ssd.Detector(ssd.default_config(W, H, max_frames_per_batch=8), ssd.transformation_for_scene(sc[0]), gpu_device)
    b = small.workspace_bytes
    small.set_single_pass(True)
    small.set_single_pass(False)
    assert small.workspace_bytes == b
    small.close()


def test_a_handle_whose_planes_do_not_fit_runs_two_passes(ssd, gpu_device, monkeypatch):
    """The planes are an optimisation: ssd_create on a device with room for the workspaces but not for the planes gives a handle
    that runs two passes - the same results - instead of failing with SSD_E_NOMEM, and says why through
    ssd_last_error.  The shortage is made with SSD_MAX_PLANE_BYTES - the bound on what a handle may take for planes, for GPUs
    shared with other tenants - set so that the SECOND of three workspaces' planes crosses it: the first one's are already
    allocated and have to be given back.  (Round 5 filled the card to its last 4 MB with up to 900 allocations instead, which
    starved whatever else ran on a shared GPU and hung on the allocator's granularity: ADVICE round 5.)"""
    W, H, n = 1024, 768, 128
    sc, buf = _batch(ssd, gpu_device, W, H, n, 51000, 51)
    trans = ssd.transformation_for_scene(sc[0])
    cfg = ssd.default_config(W, H, max_frames_per_batch=n, batches_in_flight=3)
    ref = ssd.Detector(cfg, trans, gpu_device)
    plane_bytes = ssd.plane_pool_size(n, W * H) * H * (W // 64) * 8
    rest = ref.workspace_bytes - 3 * plane_bytes                  # what the handle needs without its planes
    want = _run(ref, buf, n)
    assert ref.single_pass_stats(n)["ran"]
    ref.close()
    monkeypatch.setenv("SSD_MAX_PLANE_BYTES", str(plane_bytes + plane_bytes // 2))
    det = ssd.Detector(cfg, trans, gpu_device)                     # must not raise
    assert b"SSD_MAX_PLANE_BYTES" in ssd.lib().ssd_last_error() and b"two passes" in ssd.lib().ssd_last_error()
    assert det.workspace_bytes == rest
    assert _run(det, buf, n) == want and not det.single_pass_stats(n)["ran"]
    with pytest.raises(ssd.SsdError):                               # asked for explicitly while the memory is still not there: said loudly ..
        det.set_single_pass(True)
    assert det.workspace_bytes == rest                             # .. and the handle stays whole, on two passes
    assert _run(det, buf, n) == want and not det.single_pass_stats(n)["ran"]
    monkeypatch.delenv("SSD_MAX_PLANE_BYTES")
    det.set_single_pass(True)                                      # the memory is there again: the planes come back
    assert det.workspace_bytes == rest + 3 * plane_bytes
    assert _run(det, buf, n) == want and det.single_pass_stats(n)["ran"]
    det.close()
    buf.free()


def test_frames_the_plane_pool_cannot_serve_come_out_through_k_raster(ssd, oracle, gpu_device):
    """Planes are drawn from a pool per workspace (10 per frame of the largest batch, where a frame may ask for up to 24): a
    frame the pool cannot serve gets none and is rastered by k_raster like any frame the predictor does not cover.  The pool
    cut down to a third of what this batch draws, to nothing, and restored: results bit-equal each time, the planes zero
    afterwards, fewer frames covered while the pool is short."""
    W, H, n = 1024, 768, 96
    sc, buf = _batch(ssd, gpu_device, W, H, n, 53000, 53)
    cfg, trans = ssd.default_config(W, H, max_frames_per_batch=n), ssd.transformation_for_scene(sc[0])
    det = ssd.Detector(cfg, trans, gpu_device)
    want = _run(det, buf, n)
    full = det.single_pass_stats(n)
    assert full["ran"] and full["covered"] >= n * 9 // 10 and full["dirty_words"] == 0
    size = det.plane_pool(-1)
    assert size == ssd.plane_pool_size(n, W * H) and det.workspace_bytes <= 2.8e6 * n + 6e6
    det.single_pass(1)                        # forced: a batch k_raster did most of would switch the next ones to two passes (ssd_fetch_back)
    for planes in (full["planes"] // 3, 0):
        det.plane_pool(planes)
        assert _run(det, buf, n) == want
        st = det.single_pass_stats(n)
        assert st["ran"] and st["planes"] <= planes and st["covered"] < full["covered"] and st["dirty_words"] == 0
    det.plane_pool(-1)
    assert _run(det, buf, n) == want and det.single_pass_stats(n)["covered"] == full["covered"]
    det.plane_pool(full["planes"] // 2)
    det.enqueue(buf.ptr, n)
    res = det.fetch_list(n)
    rep = {}
    assert parity.check_batch_against_oracle(ssd, oracle, cfg, trans.constants, buf, W * H * 12, res, W, H, report=rep) == n
    assert rep.get("max_corner_err", 0.0) == 0.0
    det.close()
    buf.free()
