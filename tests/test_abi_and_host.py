"""CPU-side checks of the product: the C-ABI library loads and exports every
symbol include/mcgpu.h declares, fails loudly without a device, and the host
logic (packet sharding, result packing) is correct.  No compute call is made."""
import os
import re

import numpy as np
import pytest

import mcfost_amd.engine as eng
from mcfost_amd import distributed as D
from mcfost_amd.host import model as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "mcgpu.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mcgpu_[a-zA-Z0-9_]+)\s*\(", txt)))


def test_header_symbols_are_exported():
    import __graft_entry__ as g
    g.build_hip()   # (a no-op when the library is current; hipcc cross-compiles gfx950 without a GPU)
    lib = eng.load_library()
    decl = _declared_symbols()
    assert len(decl) >= 20
    for s in decl:
        assert hasattr(lib, s), f"{s} declared in include/mcgpu.h but not exported"
    assert sorted(eng.ABI_SYMBOLS) == decl


def test_no_cpu_fallback_without_device(small_model):
    """On a box without a GPU the engine must refuse to run, not fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(eng.McgpuError):
        eng.Engine(small_model, 1000)


def test_product_does_not_import_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "mcfost_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".f90", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "mc_oracle" not in src and "import oracle" not in src and "from oracle" not in src, f


def test_the_library_holds_no_oracle_symbol_and_needs_no_oracle_library():
    """The library's host side (host_tail.cpp: the last packets of a launch's tail on CPU threads) is the product's own
    device source compiled for the CPU -- not the CPU oracle: no symbol of the library, defined or undefined, names the
    oracle, and it links nothing under oracle/."""
    import subprocess
    import __graft_entry__ as g
    lib = g.build_hip()
    syms = subprocess.run(["nm", "-C", lib], capture_output=True, text=True).stdout + \
        subprocess.run(["nm", "-D", "-C", lib], capture_output=True, text=True).stdout
    assert len(syms) > 1000
    assert "oracle" not in syms.lower()
    assert "tail_packet" in syms   # (the host instantiations of the device source's tail_packet are in it)
    needed = subprocess.run(["readelf", "-d", lib], capture_output=True, text=True).stdout
    assert "oracle" not in needed.lower()


def test_shard_packets_partitions_the_range():
    for n in (0, 1, 7, 1000, 10 ** 9 + 7):
        for world in (1, 2, 3, 8):
            parts = [D.shard_packets(n, r, world) for r in range(world)]
            assert sum(c for _, c in parts) == n
            nxt = 0
            for first, count in parts:
                assert first == nxt
                nxt += count
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1


def test_pack_unpack_roundtrip():
    res = dict(E_abs=np.arange(5.0), sed=np.arange(24.0).reshape(2, 1, 3, 4), n_sent=np.arange(4.0),
               counters=dict(a=1, b=2))
    acc = D.pack_results(res)
    assert acc.size == 5 + 24 + 4 + 2     # ONE buffer: [E_abs | sed | n_sent | counters]
    out = D.unpack_results(acc, res)
    assert np.array_equal(out["E_abs"], res["E_abs"]) and np.array_equal(out["sed"], res["sed"])
    assert np.array_equal(out["n_sent"], res["n_sent"]) and out["counters"] == res["counters"]


def test_model_tables_are_consistent(ref41_model):
    m = ref41_model
    assert m.n_cells == 7000 and m.n_lambda == 50 and m.tab_Temp.size == 100
    assert np.all(np.diff(m.spectre_emission_cumul) >= 0) and m.spectre_emission_cumul[-1] == 1.0
    assert np.all(np.diff(m.kdB_dT_CDF, axis=1) >= 0) and np.allclose(m.kdB_dT_CDF[:, -1], 1.0)
    assert np.all(np.diff(m.log_Qcool[1:]) > 0) and m.log_Qcool[0] == -1000.0
    assert np.all(np.diff(m.prob_s11_pos.astype(float), axis=1)[:, 1:] >= -1e-7)
    assert np.allclose(m.prob_s11_pos[:, -1], 1.0, atol=1e-6) and np.all(m.prob_s11_pos[:, 0] == 0)
    assert 0 < m.albedo.min() and m.albedo.max() < 1
    # dust mass integrates back to the parameter-file value (density.f90:1976)
    mass = np.sum(m.rho_dust * m.grid["volume"]) * M.AU_TO_CM ** 3 / M.MSUN_TO_G
    assert abs(mass / m.cfg.dust_mass - 1) < 1e-12
    # kappa * kappa_factor is kappa_ext * rho * AU_to_cm (dust_prop.f90:1372)
    assert m.kappa_factor[m.extra["icell_ref"] - 1] == 1.0


def test_3d_model_has_the_baseline_cell_count():
    g = M.define_cylindrical_grid(M.ref41_3d())
    assert g["n_cells"] == 720000 and g["ntot2"] == 102 * 102 * 72


def test_more_gpus_than_the_node_has_is_an_error_not_a_number():
    """`bench.py --gpus N` started plainly on a node with fewer GPUs must fail with a message, not print a line."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64", "--packets", "1e5", "--steps", "1",
                          "--warmup", "0", "--no-cpu-baseline", "--no-extra"], capture_output=True, text=True, timeout=600,
                         env=env, cwd=root)
    assert out.returncode != 0 and "--gpus 64" in out.stderr and "GPU(s)" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
