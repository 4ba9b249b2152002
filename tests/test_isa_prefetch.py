"""The next tile's loads of K1 must be IN FLIGHT while the current tile is processed — checked in the ISA of the built library.

Round 5 found that every streaming kernel had waited for its "prefetch" right where it was issued since round 2 (the compiler copies
parts of the loaded registers at once when the loads sit behind branches; profiles/r05_k1_variants.txt (m)), and that whether it does
is the register allocator's decision at K1's budget of 96 registers: a harmless-looking change to the loop brought the copies back
((o)).  So the property is pinned here: in the gfx950 code object inside lib/libssd_hip.so (disassembled by tools/isa_prefetch.py in
under a second, no GPU), the aligned-vertex instantiations of K1 hold a group of 16-byte loads with hundreds of instructions between
them and the next `s_waitcnt vmcnt`.  k_raster / k_inquad are not held to it (their loads are still waited for at once; K4 is bound
by its instructions)."""
import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
isa = importlib.import_module("isa_prefetch")

pytestmark = pytest.mark.skipif(not (isa.tools_present() and os.path.exists(isa.DEFAULT_LIB)),
                                reason="needs the built lib/libssd_hip.so and the ROCm llvm tools")


@pytest.fixture(scope="module")
def kernels():
    return isa.disassemble()


def _one(kernels, fragment):
    names = [n for n in kernels if fragment in n]
    assert len(names) == 1, (fragment, names)
    return kernels[names[0]]


# (the last template flag, round 6: the instantiation with the rare configurations' per-point tests - launch_hist picks by configuration)
@pytest.mark.parametrize("fragment,least", [("13k_hist_planesILi1ELb0ELb0EE", 400),  # the single pass's K1, a tile a whole number of rows (XGA)
                                            ("13k_hist_planesILi1ELb0ELb1EE", 400),
                                            ("13k_hist_planesILi1ELb1ELb0EE", 400),  # the sorted strips (FHD, VGA)
                                            ("13k_hist_planesILi1ELb1ELb1EE", 400),
                                            ("6k_histILi1ELb0EE", 300),              # two passes
                                            ("6k_histILi1ELb1EE", 300)])
def test_k1_keeps_the_next_tiles_loads_in_flight(kernels, fragment, least):
    ins = _one(kernels, fragment)
    d = isa.load_distances(ins)
    assert len(d) >= 6, "expected the prologue's and the loop's three 16-byte loads"
    far = [n for _, n in d if n >= least]
    assert len(far) >= 3, ("the loop's loads are waited for %s instructions after they are issued: the prefetch is none "
                           "(see profiles/r05_k1_variants.txt (m), (o))" % [n for _, n in d])


def test_the_disassembly_is_the_shipped_kernels(kernels):
    # every instantiation the launchers can pick is in the code object
    for fragment in ("k_hist_planesILi%dELb%dELb%dEE" % (s, b, c) for s in (0, 1, 2) for b in (0, 1) for c in (0, 1)):
        _one(kernels, fragment)
    for fragment in ("6k_histILi0ELb0EE", "6k_histILi1ELb1EE", "6k_histILi2ELb0EE", "9k_predictILi1EE", "7k_peaks", "7k_quads"):
        assert any(fragment in n for n in kernels), fragment


# Round 6: the hot instantiations of K1 keep no register in scratch memory and reload no scalar from a vector register inside their
# loops (profiles/r06_kernel_resources.txt; the round's 948 -> 841 M vector instructions rest on it, and one more live value in the
# tile loop brings the spills back: DESIGN.md section 3).  The counts below are whole-kernel (prologue and epilogue included).
@pytest.mark.parametrize("fragment,lanes_moved", [("13k_hist_planesILi1ELb0ELb0EE", 64),   # XGA single pass: 44 when written
                                                   ("13k_hist_planesILi1ELb1ELb0EE", 72),   # FHD / VGA strips: 51
                                                   ("6k_histILi1ELb0EE", 8)])               # two passes: 0
def test_k1_keeps_its_registers(kernels, fragment, lanes_moved):
    ins = _one(kernels, fragment)
    assert not [l for l in ins if "scratch_" in l], "K1 spills vector registers to scratch memory"
    moved = [l for l in ins if "v_readlane_b32" in l or "v_writelane_b32" in l]
    assert len(moved) <= lanes_moved, "%d v_readlane / v_writelane: scalar registers are being spilled in K1's loops" % len(moved)
