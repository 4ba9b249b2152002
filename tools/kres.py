#!/usr/bin/env python3
"""tools/kres.py [file.s] — registers, spills, LDS and scratch of every kernel in a hipcc -S listing
(default: compiles csrc/ssd_kernels.hip for gfx950 first)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/ssd_kernels.s"
if len(sys.argv) < 2:
    subprocess.run(["hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "-S", "--cuda-device-only",
                    os.path.join(ROOT, "stair-step-detector_amd", "csrc", "ssd_kernels.hip"), "-o", path], check=True, stderr=subprocess.DEVNULL)
text = open(path).read()
for block in text.split("  - .agpr_count:")[1:]:
    def g(k):
        m = re.search(r"\." + k + r":\s+(\S+)", block)
        return m.group(1) if m else "?"
    print("%-48s sgpr %3s vgpr %3s spill(s/v) %s/%s lds %6s scratch %s" % (g("name")[:48], g("sgpr_count"), g("vgpr_count"),
          g("sgpr_spill_count"), g("vgpr_spill_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
