#!/usr/bin/env python3
"""tools/isa_prefetch.py [LIB] — are the next tile's loads of the streaming kernels in flight while the current tile is processed?

Disassembles the gfx950 code object inside libssd_hip.so (llvm-objcopy --dump-section .hip_fatbin, clang-offload-bundler
--unbundle, llvm-objdump -d: a few seconds, no GPU) and reports, for every 16-byte (12-byte) global load of the named kernels, how
many instructions lie between it and the next `s_waitcnt vmcnt`.  Round 5 found that the "prefetch" of every streaming kernel had
been waited for right where it was issued (0 - 6 instructions) since round 2: the compiler copies parts of the loaded registers at
once when the loads sit behind branches, and whether it does is decided by the register allocator (DESIGN.md section 3,
profiles/r05_k1_variants.txt (m), (o)).  tests/test_isa_prefetch.py holds the kernels that were fixed to what this prints.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
DEFAULT_LIB = os.path.join(ROOT, "stair-step-detector_amd", "lib", "libssd_hip.so")
LOAD = re.compile(r"^\s*global_load_dwordx[34]\b")
WAIT = re.compile(r"^\s*s_waitcnt\b.*vmcnt\(")
INSTR = re.compile(r"^\s+[a-z_0-9]+\b")


def tools_present():
    return all(os.path.exists(os.path.join(LLVM, t)) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump"))


def disassemble(lib=DEFAULT_LIB):
    """-> {mangled kernel name: [instruction lines]} of the gfx950 code object in `lib`"""
    tmp = tempfile.mkdtemp(prefix="ssd_isa_")
    try:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib, os.path.join(tmp, "copy.so")], check=True)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
        text = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out, name = {}, None
    for line in text.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(\S+)>:$", line)
        if m:
            name = m.group(1)
            out[name] = []
        elif name is not None and INSTR.match(line):
            out[name].append(line.split("//")[0].rstrip())
    return out


def load_distances(instrs):
    """[(index of the load, instructions until the next s_waitcnt vmcnt)] for every wide global load of one kernel"""
    res = []
    for i, ins in enumerate(instrs):
        if LOAD.match(ins):
            n = 0
            for later in instrs[i + 1:]:
                if WAIT.match(later):
                    break
                n += 1
            res.append((i, n))
    return res


def longest_flight(instrs):
    """the largest distance any wide load of the kernel has to its wait: the prefetch of the main loop, when it is one"""
    d = load_distances(instrs)
    return max((n for _, n in d), default=0)


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else DEFAULT_LIB
    kernels = disassemble(lib)
    for name in sorted(kernels):
        if not re.search(r"k_hist|k_raster|k_inquad|k_risers|k_predict", name):
            continue
        d = load_distances(kernels[name])
        short = re.sub(r"EvPK.*$", "", name)
        print("%-44s %5d instructions; wide loads -> instructions to the next vmcnt wait: %s" % (short, len(kernels[name]), " ".join(str(n) for _, n in d)))


if __name__ == "__main__":
    main()
