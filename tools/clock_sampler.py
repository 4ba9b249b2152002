#!/usr/bin/env python3
"""tools/clock_sampler.py — samples the GPU's clocks and power from sysfs in a process of its own (never touches HIP).

    python tools/clock_sampler.py OUT.jsonl [period_s] [pci_bus_id]

pci_bus_id (e.g. 0000:05:00.0): only the card at that PCI address is sampled (a host shows every GPU of the node).

Started by tools/clockstate.py BEFORE that process initialises the GPU; stops when OUT.jsonl.stop appears.
One JSON object per sample: wall-clock time, the active level of every pp_dpm_* table (the line marked `*`),
hwmon frequencies / power, gpu_busy_percent.  Sources that do not exist or are not readable are left out; the
first line of the file lists what was found.
"""
import glob
import json
import os
import re
import sys
import time


def read(path):
    try:
        with open(path) as f:
            return f.read()
    except Exception:
        return None


def active_level(text):
    """'0: 132Mhz\n1: 2100Mhz *' -> 2100.0 (MHz of the starred line), or None"""
    if not text:
        return None
    for line in text.splitlines():
        if line.rstrip().endswith("*"):
            m = re.search(r"([0-9.]+)\s*[Mm][Hh][Zz]", line)
            if m:
                return float(m.group(1))
    return None


def main():
    out = sys.argv[1]
    period = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
    want = sys.argv[3].lower() if len(sys.argv) > 3 else ""
    cards = sorted(d for d in glob.glob("/sys/class/drm/card[0-9]*") if re.fullmatch(r".*/card[0-9]+", d)
                   and os.path.exists(os.path.join(d, "device", "vendor"))
                   and (read(os.path.join(d, "device", "vendor")) or "").strip() == "0x1002")
    if want:
        mine = [c for c in cards if want in os.path.realpath(os.path.join(c, "device")).lower()]
        cards = mine or cards
    sources = {}
    for c in cards:
        dev = os.path.join(c, "device")
        name = os.path.basename(c)
        for t in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk"):
            p = os.path.join(dev, t)
            if read(p) is not None:
                sources["%s.%s" % (name, t)] = ("dpm", p)
        p = os.path.join(dev, "gpu_busy_percent")
        if read(p) is not None:
            sources["%s.busy" % name] = ("int", p)
        for hw in glob.glob(os.path.join(dev, "hwmon", "hwmon*")):
            for f in ("freq1_input", "freq2_input", "power1_average", "power1_input", "temp1_input"):
                p = os.path.join(hw, f)
                if read(p) is not None:
                    sources["%s.%s" % (name, f)] = ("int", p)
    with open(out, "w") as f:
        f.write(json.dumps({"sources": {k: v[1] for k, v in sources.items()}, "period_s": period}) + "\n")
        f.flush()
        while not os.path.exists(out + ".stop"):
            s = {"t": time.time()}
            for k, (kind, p) in sources.items():
                txt = read(p)
                if kind == "dpm":
                    v = active_level(txt)
                else:
                    try:
                        v = int(txt.strip())
                    except Exception:
                        v = None
                if v is not None:
                    s[k] = v
            f.write(json.dumps(s) + "\n")
            f.flush()
            time.sleep(period)


if __name__ == "__main__":
    main()
