#!/usr/bin/env python3
"""tools/clockstate.py — does K1's time follow what the memory system delivers, whatever state the GPU's clocks are in?

Same box, same process, interleaved for `--seconds` of continuous load: a plain read stream over the batch
(ssd_test_stream_read = tools/loadbench.hip variant C), K1 alone (ssd_enqueue_stages(HIST | PEAKS), HIP events around
the kernel), the whole pipeline (stage events) — while a SEPARATE process (tools/clock_sampler.py, sysfs only) samples
sclk / mclk / fclk / power.  Writes one JSON (default profiles/r03_clockstate.json): every round with its times, the
ratio k1 / plain stream, and the clock samples that fall into the round.

    python tools/clockstate.py [--seconds 45] [--frames 1024] [--out profiles/r03_clockstate.json]
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=45.0)
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--idle", type=float, default=5.0, help="seconds of idle in the middle of the run (does the state fall back?)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r03_clockstate.json"), help="(copy it to profiles/ afterwards)")
    args = ap.parse_args()

    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    samples_path = os.path.join(ROOT, "gpurun_out", "clock_samples.jsonl")
    for p in (samples_path, samples_path + ".stop"):
        if os.path.exists(p):
            os.remove(p)
    # Which card is ours?  A child asks HIP for device 0's PCI address and exits; the sampler — which must never share a
    # process with HIP — is started on that card before this process touches the GPU.
    probe = subprocess.run([sys.executable, "-c",
                            "import ctypes; h = ctypes.CDLL('libamdhip64.so'); b = ctypes.create_string_buffer(64); "
                            "rc = h.hipDeviceGetPCIBusId(b, 64, 0); print(b.value.decode() if rc == 0 else '')"],
                           capture_output=True, text=True)
    bus = probe.stdout.strip().splitlines()[-1] if probe.stdout.strip() else ""
    sampler = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "clock_sampler.py"), samples_path, "0.02", bus])
    time.sleep(1.0)                         # a second of idle clocks on record

    ssd = importlib.import_module("stair-step-detector_amd")
    import scenes
    W, H, F = 1024, 768, args.frames
    sc = scenes.batch_scenes(ssd, W, H, F, base_seed=100000, rng_seed=1000)
    trans = ssd.transformation_for_scene(sc[0])
    cfg = ssd.default_config(W, H, max_frames_per_batch=F)
    nbytes = W * H * 12 * F
    buf = ssd.DeviceBuffer(nbytes, 0)
    ssd.synth_device(sc, buf.ptr, device=0)
    det = ssd.Detector(cfg, trans, 0)
    det.set_timing(True)
    lib = ssd.lib()
    lib.ssd_device_sync(0)

    rounds = []
    t_start = time.time()
    idled = False
    while time.time() - t_start < args.seconds:
        if not idled and args.idle > 0 and time.time() - t_start > args.seconds * 0.6:
            time.sleep(args.idle)
            idled = True
            rounds.append({"t0": time.time(), "event": "resumed after %.1f s idle" % args.idle})
        r = {"t0": time.time()}
        r["stream_ms"] = ssd.stream_read_ms(buf.ptr, nbytes, reps=10)
        k1 = []
        for _ in range(10):
            det.enqueue(buf.ptr, F, stages=ssd.STAGE_HIST | ssd.STAGE_PEAKS)
            lib.ssd_device_sync(0)
            k1.append(det.stage_times_ms()["hist"])
        r["k1_alone_ms"] = sum(k1) / len(k1)
        full = {k: 0.0 for k in ssd.STAGE_NAMES}
        for _ in range(5):
            det.enqueue(buf.ptr, F)
            det.fetch(F)
            for k, v in det.stage_times_ms().items():
                full[k] += v / 5
        r["pipeline_stage_ms"] = full
        r["stream_ms_after"] = ssd.stream_read_ms(buf.ptr, nbytes, reps=10)
        r["t1"] = time.time()
        r["stream_GBps"] = nbytes / (0.5 * (r["stream_ms"] + r["stream_ms_after"])) / 1e6
        r["k1_GBps"] = nbytes / r["k1_alone_ms"] / 1e6
        r["k1_over_plain_stream"] = 0.5 * (r["stream_ms"] + r["stream_ms_after"]) / r["k1_alone_ms"]
        rounds.append(r)
    det.close()
    buf.free()

    open(samples_path + ".stop", "w").close()
    sampler.wait(timeout=10)
    samples = [json.loads(line) for line in open(samples_path) if line.strip()]
    header, samples = samples[0], samples[1:]
    keys = sorted({k for s in samples for k in s if k != "t"})

    def mean_in(t0, t1, key):
        v = [s[key] for s in samples if t0 <= s["t"] <= t1 and key in s]
        return sum(v) / len(v) if v else None

    for r in rounds:
        if "t1" in r:
            r["clocks"] = {k: mean_in(r["t0"], r["t1"], k) for k in keys}
    idle = {k: mean_in(0, t_start - 0.2, k) for k in keys}
    timed = [r for r in rounds if "t1" in r]
    ratios = [r["k1_over_plain_stream"] for r in timed]
    out = {
        "what": "same box, one process, interleaved for %.0f s: plain read stream (ssd_test_stream_read), K1 alone, whole pipeline; "
                "clocks sampled every 50 ms from sysfs by a separate process (tools/clock_sampler.py)" % args.seconds,
        "frames": F, "bytes": nbytes, "pci_bus_id": bus, "clock_sources": header.get("sources"), "idle_before_load": idle,
        "summary": {
            "rounds": len(timed),
            "k1_alone_ms_min_max": [min(r["k1_alone_ms"] for r in timed), max(r["k1_alone_ms"] for r in timed)],
            "stream_GBps_min_max": [min(r["stream_GBps"] for r in timed), max(r["stream_GBps"] for r in timed)],
            "k1_over_plain_stream_min_max_mean": [min(ratios), max(ratios), sum(ratios) / len(ratios)],
        },
        "rounds": rounds,
    }
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out["summary"]))
    print("clock sources:", json.dumps(header.get("sources")))
    for r in timed[:: max(1, len(timed) // 12)]:
        print("t=%6.1f s  stream %.0f GB/s  K1 %.3f ms (%.0f GB/s)  ratio %.3f  K2 %.3f K4 %.3f  clocks %s"
              % (r["t0"] - t_start, r["stream_GBps"], r["k1_alone_ms"], r["k1_GBps"], r["k1_over_plain_stream"],
                 r["pipeline_stage_ms"]["raster"], r["pipeline_stage_ms"]["inquad"],
                 {k: (round(v) if v is not None else None) for k, v in r["clocks"].items()}))


if __name__ == "__main__":
    main()
