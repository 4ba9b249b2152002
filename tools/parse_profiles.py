#!/usr/bin/env python3
"""tools/parse_profiles.py TAG — condenses gpurun_out/prof_TAG (rocprofv3 CSVs) into profiles/TAG_*.

  profiles/TAG_kernel_stats.csv     rocprofv3 --kernel-trace --stats summary (per-kernel calls / average ns)
  profiles/TAG_hbm_traffic.json     per-kernel FETCH_SIZE / WRITE_SIZE per launch, corrected as
                                    MI355X_MICROARCH.md prescribes: counters are KiB; on gfx950 FETCH_SIZE reads
                                    exactly 1/2 of a wide (16 B/lane) coalesced streaming read -> x2 for the
                                    streaming kernels (k_hist, k_raster, k_inquad); WRITE_SIZE is exact.
  profiles/pmc_k_hist.json          what bench.py reads for roofline.traffic (HBM bytes per k_hist launch)
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STREAMING = ("k_hist", "k_predict", "k_raster", "k_inquad", "k_stream_read")      # 16-byte loads per lane ("k_hist" also names k_hist_planes)
NOT_PIPELINE = ("synth", "k_stream_read")          # frame generator; bench.py's plain read stream beside K1


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def load_stamp(src):
    """the stamp the run wrote next to its outputs (tools/r06_final.sh: git revision of the build, sha256 of the libssd_hip.so the
    passes loaded, checked on the GPU box against build/STAMP.json), or None for runs of earlier rounds"""
    p = os.path.join(src, "stamp.json")
    return json.load(open(p)) if os.path.exists(p) else None


def dump(obj, path, stamp):
    if stamp is not None:
        obj = dict(obj)
        obj["stamp"] = stamp
    json.dump(obj, open(path, "w"), indent=1)


def newest(paths):
    """the most recent of the files a pattern found, as a one-element list (a tag collected twice leaves both runs' directories)"""
    return [max(paths, key=os.path.getmtime)] if paths else []


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    stamp = load_stamp(src)
    if stamp is not None:
        json.dump(stamp, open(os.path.join(dst, tag + "_stamp.json"), "w"), indent=1)       # for the csv summaries, which cannot carry it
    stats = newest(glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv")))
    if stats:
        shutil.copy(stats[0], os.path.join(dst, tag + "_kernel_stats.csv"))
        # what bench.py reports as roofline.frac_profiles (round 6): K1's rocprofv3 average of this pass, and on which GPU
        for r in csv.DictReader(open(stats[0])):
            if "k_hist_planes" in r["Name"]:
                avg_ms = float(r["AverageNs"]) / 1e6
                gpu = None
                try:
                    b = json.loads(open(os.path.join(src, "bench.json")).read().strip().split("\n")[-1])
                    gpu = {k: b["devices"][0].get(k) for k in ("uuid", "pci_bus_id", "name")}
                except Exception:
                    pass
                alg = 12.0 * 1024 * 768 * 1024
                keep = {}
                try:        # further passes recorded by hand for the same library (tools/rocprof_states.sh) stay with it
                    old = json.load(open(os.path.join(dst, "k1_rocprof.json")))
                    if stamp is not None and (old.get("stamp") or {}).get("lib_sha256") == stamp.get("lib_sha256") and "other_rocprof_passes_same_library" in old:
                        keep = {"other_rocprof_passes_same_library": old["other_rocprof_passes_same_library"]}
                except Exception:
                    pass
                dump({**keep, "kernel": r["Name"].split("(")[0].replace("void ", ""), "calls": int(r["Calls"]), "avg_launch_ms": avg_ms,
                      "algorithmic_bytes_per_launch": alg, "frac": alg / (avg_ms * 1e-3) / 8e12, "gpu": gpu,
                      "summary": "profiles/%s_kernel_stats.csv (rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --batches-in-flight 1 ...)" % tag},
                     os.path.join(dst, "k1_rocprof.json"), stamp)
                break
    out = {"note": "per launch = per batch of 1024 frames of 1024x768 points; bytes", "kernels": {}}
    fetch = newest(glob.glob(os.path.join(src, "fetch", "*", "*counter_collection.csv")))
    write = newest(glob.glob(os.path.join(src, "write", "*", "*counter_collection.csv")))
    f = per_kernel(fetch[0], "FETCH_SIZE") if fetch else {}
    w = per_kernel(write[0], "WRITE_SIZE") if write else {}
    for k in sorted(set(f) | set(w)):
        if not k.startswith("ssd::"):
            continue
        corr = 2.0 if any(s in k for s in STREAMING) else 1.0
        out["kernels"][k] = {"FETCH_SIZE_KiB_raw": f.get(k), "fetch_correction": corr,
                             "hbm_read_bytes": None if k not in f else f[k] * 1024 * corr,
                             "WRITE_SIZE_KiB_raw": w.get(k), "hbm_write_bytes": None if k not in w else w[k] * 1024}
    dump(out, os.path.join(dst, tag + "_hbm_traffic.json"), stamp)
    for k, v in out["kernels"].items():
        if "k_hist" in k and v["hbm_read_bytes"]:
            total = v["hbm_read_bytes"] + (v["hbm_write_bytes"] or 0.0)
            dump({"source": "profiles/%s_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)" % tag,
                  "hbm_bytes_per_launch_at_1024x768x1024": total,
                  "algorithmic_bytes_per_launch": 12.0 * 1024 * 768 * 1024,
                  "ratio": total / (12.0 * 1024 * 768 * 1024)}, os.path.join(dst, "pmc_k_hist.json"), stamp)
    moved_r = sum((v["hbm_read_bytes"] or 0.0) for k, v in out["kernels"].items() if not any(n in k for n in NOT_PIPELINE))
    moved_w = sum((v["hbm_write_bytes"] or 0.0) for k, v in out["kernels"].items() if not any(n in k for n in NOT_PIPELINE))
    if moved_r:
        dump({"source": "profiles/%s_hbm_traffic.json (all kernels of one pass over 1024 XGA frames; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)" % tag,
              "hbm_read_bytes": moved_r, "hbm_write_bytes": moved_w, "algorithmic_bytes": 12.0 * 1024 * 768 * 1024,
              "bytes_moved_over_algorithmic": (moved_r + moved_w) / (12.0 * 1024 * 768 * 1024)},
             os.path.join(dst, "pmc_pipeline.json"), stamp)
    for name, to in (("bench.json", "_bench.json"), ("bench_fhd.json", "_bench_fhd_stress.json"), ("bench_depth16.json", "_bench_depth16.json"),
                     ("hostfed.json", "_hostfed.json"), ("latency.txt", "_latency.txt")):
        b = os.path.join(src, name)
        if os.path.exists(b) and os.path.getsize(b):
            shutil.copy(b, os.path.join(dst, tag + to))
    stats = newest(glob.glob(os.path.join(src, "trace_fhd", "*", "*kernel_stats.csv")))
    if stats:
        shutil.copy(stats[0], os.path.join(dst, tag + "_kernel_stats_fhd_stress.csv"))
    # BASELINE configs[4]: HBM read traffic and bandwidth fraction of the streaming kernels on the FHD stress batch (256 frames)
    fetch_fhd = newest(glob.glob(os.path.join(src, "fetch_fhd", "*", "*counter_collection.csv")))
    if fetch_fhd and stats:
        ff = per_kernel(fetch_fhd[0], "FETCH_SIZE")
        avg_ns = {}
        for r in csv.DictReader(open(stats[0])):
            avg_ns[r["Name"].split("(")[0].replace("void ", "")] = float(r["AverageNs"])
        alg = 12.0 * 1920 * 1080 * 256
        o = {"note": "FHD stress (BASELINE configs[4]): 256 frames of 1920x1080 per launch; FETCH_SIZE KiB x 1024 x 2 for the streaming kernels "
                     "(MI355X_MICROARCH.md: the counter reads half of a 16-B-per-lane stream); durations from the --stats pass of the same command",
             "algorithmic_bytes_per_launch": alg, "kernels": {}}
        for k, v in sorted(ff.items()):
            if not k.startswith("ssd::") or any(n in k for n in NOT_PIPELINE):
                continue
            corr = 2.0 if any(x in k for x in STREAMING) else 1.0
            rd = v * 1024 * corr
            e = {"hbm_read_bytes": rd, "avg_ms": None if k not in avg_ns else avg_ns[k] / 1e6}
            if k in avg_ns and avg_ns[k] > 0:
                e["hbm_read_GBps"] = rd / (avg_ns[k] * 1e-9) / 1e9
                e["frac_of_8TBps_peak"] = e["hbm_read_GBps"] / 8000.0
            if "k_hist" in k and k in avg_ns:
                e["algorithmic_frac_of_peak"] = alg / (avg_ns[k] * 1e-9) / 1e9 / 8000.0
            o["kernels"][k] = e
        dump(o, os.path.join(dst, tag + "_hbm_traffic_fhd_stress.json"), stamp)
    # instruction / issue counters of a tools/pmc.sh run (optional second argument: its tag)
    if len(sys.argv) > 2:
        agg = collections.defaultdict(dict)
        for f in glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_" + sys.argv[2], "*", "*", "*counter_collection.csv")):
            per = collections.defaultdict(lambda: collections.defaultdict(list))
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                if k.startswith("ssd::k_") and not any(n in k for n in NOT_PIPELINE):
                    per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k in per:
                for c, v in per[k].items():
                    agg[k][c] = round(sum(v) / len(v), 3)
        issue = {"note": "rocprofv3 --kernel-trace --pmc <set> -- python3 bench.py --steps 2 --warmup 1 --no-cpu (one pass per counter set, "
                         "tools/pmc.sh); per launch = 1024 XGA frames", "kernels": {k: dict(sorted(v.items())) for k, v in sorted(agg.items())}}
        dump(issue, os.path.join(dst, tag + "_pmc_issue.json"), stamp)
        issue["source"] = "profiles/%s_pmc_issue.json" % tag
        dump(issue, os.path.join(dst, "pmc_issue.json"), stamp)      # what bench.py reads for `floors`
    print(open(os.path.join(dst, tag + "_hbm_traffic.json")).read()[:1500])


if __name__ == "__main__":
    main()
