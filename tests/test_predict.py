"""The single pass's predictor table (csrc/ssd_predict.h, host statement; k_predict computes the same on the device - held against
it in tests/test_gpu_single_pass.py).  No GPU: properties on constructed sample histograms, and that a sample as good as the
complete histogram covers every step plateau the oracle finds on the golden scenes."""
import numpy as np
import pytest

import scenes

SAMPLE = 16          # kSpecSample: a sample count stands for 16 points
NONE = 255


def _table(ssd, counts, n_bins=121, min_height=15, sabotage=0):
    s = np.zeros(ssd.MAX_BINS, dtype=np.uint32)
    for b, c in counts.items():
        s[b] = c
    return ssd.predict_table_host(s, n_bins, min_height, sabotage)


def test_a_tread_in_two_bins_gets_one_plane_for_both(ssd):
    plane, n = _table(ssd, {29: 3000, 30: 2500, 31: 100, 28: 120})
    assert n == 1 and plane[29] == 0 and plane[30] == 0 and (plane[np.arange(128) < 29] == NONE).all() and plane[31] == NONE and plane[28] == NONE
    plane, n = _table(ssd, {29: 2500, 30: 3000, 31: 100, 28: 120})          # the peak on the other side
    assert n == 1 and plane[29] == 0 and plane[30] == 0 and plane[31] == NONE


def test_a_tread_in_one_bin_with_like_neighbours_gets_three_planes(ssd):
    plane, n = _table(ssd, {48: 150, 49: 4500, 50: 160})
    assert n == 3 and list(plane[48:51]) == [0, 1, 2] and plane[47] == NONE and plane[51] == NONE


def test_two_almost_equal_bins_are_one_peak(ssd):
    """FHD stress: 51 566 and 52 912 points in neighbouring bins - both pass the peak filter, the fuller one is the candidate"""
    plane, n = _table(ssd, {20: 12, 21: 3223, 22: 3307, 23: 125})
    assert n == 1 and plane[21] == 0 and plane[22] == 0 and plane[20] == NONE and plane[23] == NONE


def test_flat_background_small_peaks_and_bins_below_min_height_get_nothing(ssd):
    flat = {b: 150 + (b % 3) for b in range(12, 121)}                       # a wall behind the stairs: every bin up to the last one
    assert _table(ssd, flat)[1] == 0
    assert _table(ssd, {40: 70, 39: 2, 41: 3})[1] == 0                      # 1120 points scaled up: below 1200
    assert _table(ssd, {40: 76, 39: 2, 41: 3})[1] >= 1
    assert _table(ssd, {9: 6000, 10: 6200, 11: 100})[1] == 0                # the ground: below min_height
    assert _table(ssd, {15: 3000, 14: 10, 16: 12})[1] == 3                  # a peak AT min_height may take the bin below it
    assert _table(ssd, {119: 10, 120: 3000}, n_bins=121)[1] == 0            # the last bin is never a peak (findPeaks looks ahead)


def test_at_most_eight_peaks_the_fullest_and_never_more_planes_than_there_are(ssd):
    counts = {}
    for k in range(12):                                                      # twelve lone peaks, fuller with k
        counts[20 + 8 * k] = 500 + 100 * k
    plane, n = _table(ssd, counts)
    chosen = [b for b in counts if plane[b] != NONE]
    assert sorted(chosen) == sorted(20 + 8 * k for k in range(4, 12))       # the eight fullest
    assert n == 24 == ssd.MAX_PLANES and plane.max(initial=0, where=plane != NONE) == 23
    used = plane[plane != NONE]
    assert list(np.unique(used)) == list(range(24)) and (np.diff(used.astype(int)) >= 0).all()     # planes ascend with the bins


def test_sabotage_moves_or_empties_the_table(ssd):
    base, n = _table(ssd, {29: 3000, 30: 2500, 48: 150, 49: 4500, 50: 160})
    moved, n1 = _table(ssd, {29: 3000, 30: 2500, 48: 150, 49: 4500, 50: 160}, sabotage=1)
    assert n1 == n == 4 and (moved[3:] == base[:-3]).all() and (moved[:3] == NONE).all()
    assert _table(ssd, {29: 3000, 30: 2500}, sabotage=2)[1] == 0


@pytest.mark.parametrize("name", ["xga_config1", "xga_3steps_noise2mm", "xga_8steps_outliers", "xga_bin_boundary", "xga_roll3", "vga_3steps_clean", "vga_8steps_outliers", "vga_yaw_outliers", "fhd_config5", "ragged_1100x700_outliers"])
def test_a_faithful_sample_covers_the_oracles_step_plateaus(ssd, oracle, name):
    """The complete histogram of a golden scene divided by 16 as the sample: every bin of every step plateau the oracle finds must
    have a plane (or no points), and no plane of a plateau may hold a bin outside it - k_peaks' test, restated."""
    import oracle_binding as ob
    sc = scenes.make(ssd, name)
    xyz = ssd.synth_host([sc])[0]
    cfg = ssd.default_config(sc.width, sc.height)
    trans = ssd.transformation_for_scene(sc)
    res, *_ = oracle.process(ob.to_oracle_config(cfg), ob.to_oracle_calibration(trans.constants), xyz, images=0, ground_images=False)
    hist = np.array(res.hist[:ssd.MAX_BINS], dtype=np.uint32)
    plane, n = ssd.predict_table_host(hist // SAMPLE, res.n_bins, res.min_height)
    assert n <= ssd.MAX_PLANES
    consumed, steps = -1, 0
    for i in range(res.n_plateaus):
        p = res.plateaus[i]
        lo, hi = max(p.bin_lo, consumed + 1), p.bin_hi
        consumed = max(consumed, hi)
        if not p.is_step or p.n_points < 4000:           # a plateau the reference's filter passes by a hair need not show in a sample
            continue
        steps += 1
        mine = set()
        for b in range(lo, hi + 1):
            assert plane[b] != NONE or hist[b] == 0, (name, i, b)
            if plane[b] != NONE:
                mine.add(int(plane[b]))
        for b in (lo - 1, hi + 1):
            if 0 <= b < res.n_bins and plane[b] != NONE and int(plane[b]) in mine:
                assert hist[b] == 0, (name, i, b)
    assert steps >= 1
