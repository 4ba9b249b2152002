"""Named synthetic scenes shared by the CPU and GPU tests (SURVEY.md section 8(c) fixture set, BASELINE.json configs)."""

RES = {"vga": (640, 480), "xga": (1024, 768), "fhd": (1920, 1080),
       # ragged: width not a multiple of 64 (partial last image word), point count not a multiple of the 1024-point
       # block tile / of 4 (12-byte load path, partial last tile)
       "r600": (600, 450), "r427": (427, 321), "r1100": (1100, 700)}


def scene_params():
    """name -> (resolution key, make_scene kwargs)."""
    p = {}
    for r in ("vga", "xga", "fhd"):
        p["%s_3steps_clean" % r] = (r, dict(n_steps=3, sigma=0.0, seed=1))
        p["%s_3steps_noise2mm" % r] = (r, dict(n_steps=3, sigma=0.002, seed=2))
    # BASELINE.json config 1/2: one XGA frame, 3 steps, sigma 1 mm
    p["xga_config1"] = ("xga", dict(n_steps=3, sigma=0.001, seed=12345))
    # BASELINE.json config 5: FHD, 8 noisy steps + 5 % outliers
    p["fhd_config5"] = ("fhd", dict(n_steps=8, sigma=0.002, seed=5, outlier_frac=0.05, tread=0.14, rise=0.12,
                                    first_riser_y=0.15, cam_height=1.4, pitch_deg=55.0))
    p["xga_8steps_outliers"] = ("xga", dict(n_steps=8, sigma=0.002, seed=6, outlier_frac=0.05, tread=0.14, rise=0.12,
                                            first_riser_y=0.15, cam_height=1.4, pitch_deg=55.0))
    p["vga_8steps_outliers"] = ("vga", dict(n_steps=8, sigma=0.002, seed=7, outlier_frac=0.05, tread=0.14, rise=0.12,
                                            first_riser_y=0.15, cam_height=1.4, pitch_deg=55.0))
    p["xga_no_stairs"] = ("xga", dict(n_steps=0, sigma=0.001, seed=8))
    p["vga_empty"] = ("vga", dict(n_steps=3, sigma=0.001, seed=9, invalid_frac=1.0))
    p["xga_bin_boundary"] = ("xga", dict(n_steps=3, sigma=0.0005, seed=10, rise=0.2, tread=0.25))
    p["xga_narrow"] = ("xga", dict(n_steps=3, sigma=0.001, seed=11, stair_width=0.5))
    p["xga_wide"] = ("xga", dict(n_steps=3, sigma=0.001, seed=12, stair_width=1.6))
    p["xga_yaw_p8"] = ("xga", dict(n_steps=3, sigma=0.001, seed=13, yaw_deg=8.0))
    p["xga_yaw_m10"] = ("xga", dict(n_steps=3, sigma=0.0015, seed=14, yaw_deg=-10.0))
    p["xga_roll3"] = ("xga", dict(n_steps=3, sigma=0.001, seed=15, roll_deg=3.0))
    p["xga_invalid10"] = ("xga", dict(n_steps=3, sigma=0.001, seed=16, invalid_frac=0.10))
    p["vga_yaw_outliers"] = ("vga", dict(n_steps=4, sigma=0.003, seed=17, yaw_deg=6.0, outlier_frac=0.02, rise=0.15, tread=0.26))
    p["xga_low_camera"] = ("xga", dict(n_steps=2, sigma=0.001, seed=18, cam_height=0.8, pitch_deg=40.0, first_riser_y=0.6))
    # strong yaw: quadrilaterals far from axis-aligned (quirk Q9 territory); at 40-50 degrees the reference's
    # QuadrilateralTest constructor throws (quirk Q7 / quadrilateralTest.cpp:283-288): the frame has no line
    p["xga_yaw20"] = ("xga", dict(n_steps=3, sigma=0.001, seed=20, yaw_deg=20.0))
    p["vga_yaw30_narrow"] = ("vga", dict(n_steps=3, sigma=0.002, seed=100, yaw_deg=30.0, stair_width=0.5))
    p["vga_yaw40_wide_throws"] = ("vga", dict(n_steps=3, sigma=0.002, seed=100, yaw_deg=40.0, stair_width=1.1))
    p["vga_yaw50_throws"] = ("vga", dict(n_steps=3, sigma=0.002, seed=101, yaw_deg=50.0, stair_width=0.8))
    p["ragged_600x450"] = ("r600", dict(n_steps=3, sigma=0.001, seed=21))
    p["ragged_427x321_yaw"] = ("r427", dict(n_steps=3, sigma=0.001, seed=22, yaw_deg=5.0))
    p["ragged_1100x700_outliers"] = ("r1100", dict(n_steps=4, sigma=0.002, seed=23, outlier_frac=0.03, rise=0.15, tread=0.26))
    # found by tools/fuzz.py (seed 20261003): an outlier pixel on image row 1 / H-2 of a scan column — the 3x3 closing
    # also lights the border pixel next to it (the erosion ignores out-of-image neighbours), i.e. a pixel OUTSIDE
    # the bounding box of the raw bits
    p["fuzz_border_closing_0"] = ("vga", dict(n_steps=7, seed=753523895, cam_height=1.136752425315668, pitch_deg=38.99284936887012, roll_deg=2.1667327820945097, first_riser_y=0.4482937595916664, tread=0.27404233006301326, rise=0.08003462791607925, stair_width=0.9418650935580659, yaw_deg=1.4481814211961677, sigma=0.0028683304688289285, outlier_frac=0.15, invalid_frac=0.0))
    p["fuzz_border_closing_1"] = ("r600", dict(n_steps=5, seed=80944533, cam_height=1.263229686152806, pitch_deg=39.68055021127778, roll_deg=1.3312143909189542, first_riser_y=0.2255171024398212, tread=0.34592134937750907, rise=0.10170513268147316, stair_width=1.2994579211897062, yaw_deg=-3.5375798031106402, sigma=0.0029247316800371462, outlier_frac=0.05, invalid_frac=0.0))
    p["xga_2steps_deep"] = ("xga", dict(n_steps=2, sigma=0.002, seed=19, tread=0.4, rise=0.19, first_riser_y=0.35))
    return p


def make(ssd, name):
    r, kw = scene_params()[name]
    w, h = RES[r]
    return ssd.make_scene(w, h, **kw)


def batch_scenes(ssd, width, height, n, base_seed=1000, rng_seed=7):
    """BASELINE.json config 3: randomised rise / tread / yaw / noise, K = 3."""
    import numpy as np
    rng = np.random.default_rng(rng_seed)
    out = []
    for i in range(n):
        out.append(ssd.make_scene(width, height, n_steps=3, seed=base_seed + i,
                                  rise=float(rng.uniform(0.14, 0.20)), tread=float(rng.uniform(0.25, 0.32)),
                                  yaw_deg=float(rng.uniform(-10.0, 10.0)), sigma=float(rng.uniform(0.0005, 0.003))))
    return out


def fhd_stress_scenes(ssd, n, base_seed=9000):
    """BASELINE.json config 5: 1920x1080, 8 noisy steps + 5 % outliers."""
    return [ssd.make_scene(1920, 1080, n_steps=8, sigma=0.002, seed=base_seed + i, outlier_frac=0.05, tread=0.14, rise=0.12,
                           first_riser_y=0.15, cam_height=1.4, pitch_deg=55.0) for i in range(n)]
