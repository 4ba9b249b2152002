#!/usr/bin/env python3
"""tools/kres2.py — registers, spills, LDS and scratch of every kernel, from the ISA text of ssd_kernels.hip
(hipcc --offload-device-only -S; no GPU needed).  usage: python tools/kres2.py [extra hipcc flags...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "stair-step-detector_amd", "csrc", "ssd_kernels.hip")
out = os.path.join(tempfile.gettempdir(), "ssd_kernels.s")
subprocess.run(["hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "-Wno-unused-function", "-w",
                "--offload-device-only", "-S", "-x", "hip", src, "-o", out] + sys.argv[1:], check=True)
s = open(out).read()
meta = s[s.index("amdhsa.kernels:"):]
for blk in meta.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("void ssd::", "")
    print("%-28s sgpr %3s (spill %3s)  vgpr %3s (spill %2s)  lds %6s  scratch %4s" % (name, g("sgpr_count"), g("sgpr_spill_count"), g("vgpr_count"),
          g("vgpr_spill_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
print("ISA text:", out)
