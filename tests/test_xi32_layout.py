"""The packed default-real xI_scatt layout (mcfost_amd/csrc/mc_xi32.hip.h) on the CPU: every value a deposit can reach
has its own place inside the sub-bin, a packet's deposits touch the lines the layout says they touch, the arrangement is
the one with fewer lines, and the Python mirror the bench's accounting uses agrees with the header."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def lib():
    src = os.path.join(HERE, "emu", "xi32_probe.cpp")
    so = os.path.join(HERE, "emu", "libxi32_probe.so")
    hdr = os.path.join(HERE, "..", "mcfost_amd", "csrc", "mc_xi32.hip.h")
    if not os.path.exists(so) or max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(so):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-o", so, src])
    l = C.CDLL(so)
    l.xi32_probe_layout.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    l.xi32_probe_offset.argtypes = [C.c_int] * 5
    l.xi32_probe_offset.restype = C.c_int
    return l


def _layout(lib, nRT, pola, contrib):
    out = (C.c_int * 11)()
    lib.xi32_probe_layout(nRT, int(pola), int(contrib), out)
    return dict(zip(("binf", "nA", "sA", "oS", "sS", "oT", "sT", "split", "sum_I", "lines", "rowf"), out))


def test_layout_invariants_and_python_mirror(lib):
    from mcfost_amd.engine import xi32_layout
    for nRT in list(range(1, 41)) + [64, 100]:
        for pola in (False, True):
            for contrib in (False, True):
                L = _layout(lib, nRT, pola, contrib)
                nS = 4 if pola else 1
                ntf = nS + (4 if contrib else 0)
                assert L["binf"] % 16 == 0 and L["rowf"] % 4 == 0 and L["sum_I"] == int(contrib)
                # a place per reachable value, no two alike, all inside the sub-bin; I is the sum of the origins with contributions
                places = {}
                for q in range(nRT):
                    for t in range(ntf):
                        o = lib.xi32_probe_offset(nRT, int(pola), int(contrib), q, t)
                        reachable = t < nS or t in (nS + 1, nS + 3)
                        if contrib and t == 0:
                            assert o == -2
                        elif reachable:
                            assert 0 <= o < L["binf"] and o not in places, (nRT, pola, contrib, q, t, o)
                            places[o] = (q, t)
                        else:
                            assert o == -1       # direct light: never deposited by the Monte Carlo
                # the lines a stellar / a thermal packet's deposits touch
                def lines_of(star):
                    ls = set()
                    for o, (q, t) in places.items():
                        if t < nS or (t == nS + 1 and star) or (t == nS + 3 and not star):
                            ls.add(o // 16)
                    return len(ls)
                touched = max(lines_of(True), lines_of(False)) if contrib else lines_of(True)
                assert touched == L["lines"], (nRT, pola, contrib, touched, L)
                # ... and no fewer than the values of one packet occupy
                per_packet = nRT * (nS if not contrib else nS)     # (n_Stokes - 1 + one origin with contributions)
                assert lines_of(True) >= (per_packet * 4 + 63) // 64
                # the split arrangement only where it touches fewer lines than the interleaved one would
                if contrib:
                    inter = (nRT * (nS + 1) + 15) // 16
                    assert (lines_of(True) + lines_of(False) < 2 * inter) == bool(L["split"]) or not L["split"]
                    if L["split"]:
                        assert lines_of(True) + lines_of(False) < 2 * inter
                # the flights' weight rows hold the stellar image (split) or the whole sub-bin
                assert L["rowf"] == (L["binf"] if not L["split"] else (nRT * nS + 3) // 4 * 4)
                # the mirror in engine.py
                m = xi32_layout(nRT, pola, contrib)
                assert (m["binf"], m["lines_touched"], m["split"]) == (L["binf"], L["lines"], bool(L["split"])), (nRT, pola, contrib)


def test_ten_observers_are_three_lines(lib):
    """BASELINE config 2's ten inclinations with Stokes tracking and contributions: 160 bytes per packet and crossing,
    three 64-byte lines for a stellar and for a thermal packet alike."""
    L = _layout(lib, 10, True, True)
    assert L["split"] == 1 and L["binf"] == 64 and L["lines"] == 3 and L["rowf"] == 40
    L3 = _layout(lib, 3, True, True)       # ref4.1.para's own three: fifteen values, one line
    assert L3["split"] == 0 and L3["binf"] == 16 and L3["lines"] == 1
