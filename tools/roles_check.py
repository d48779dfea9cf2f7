"""Correctness and timing of the opt-in role schedule (MCGPU_ROLES=<flyer waves>, mc_roles.hip.h) against the
default kernel: same packets, same counters, same sums."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "diag":  # the -DMCGPU_COUNT_ITERS build, chosen before the engine loads
    os.environ["MCGPU_LIB"] = os.path.join(ROOT, "mcfost_amd/csrc/variants/lib_iters.so")
import numpy as np
from mcfost_amd.engine import Engine
from mcfost_amd.host import model as M

def run(m, n, roles, k_short=2, fly_iters=16, fly_idle=16, emit_qmax=1 << 20, emit_min=1, **kw):
    os.environ.pop("MCGPU_ROLES", None)
    if roles is not None:
        os.environ["MCGPU_ROLES"] = str(roles)
        os.environ["MCGPU_K_SHORT"] = str(k_short)
        os.environ["MCGPU_FLY_ITERS"] = str(fly_iters)
        os.environ["MCGPU_FLY_IDLE"] = str(fly_idle)
        os.environ["MCGPU_EMIT_QMAX"] = str(emit_qmax)
        os.environ["MCGPU_EMIT_MIN"] = str(emit_min)
    e = Engine(m, n)
    prior = kw.pop("prior", None)
    t = time.perf_counter()
    r = e.run_thermal(n, seed=7, frozen=prior is not None, E_prior=prior)
    r["wall"] = time.perf_counter() - t
    e.close()
    os.environ.pop("MCGPU_ROLES", None)
    return r

stage = sys.argv[1] if len(sys.argv) > 1 else "small"
if stage == "diag":  # needs mcfost_amd/csrc/variants/lib_iters.so (-DMCGPU_COUNT_ITERS)
    m = M.build_model(M.ref41())
    n = 20_000_000
    for roles, ks, fi, idle in ((200, 2, 16, 16), (5, 2, 64, 65)):
        r = run(m, n, roles, ks, fi, idle)
        c = r["counters"]  # carry the schedule's statistics in this build (mc_roles.hip.h)
        srv_rounds, fly_rounds = float(c["flights"]), float(c["escaped"])
        print("roles", roles, "ms %.1f" % r["kernel_ms"],
              "| per server round: %.1f lanes arrive with a flight, %.1f hand it over, %.1f pop a waiting packet, %.1f stay "
              "empty after the exchange, %.2f crossing iterations" % (c["packets"] / srv_rounds, c["crossings"] / srv_rounds,
              c["killed_star"] / srv_rounds, c["scatterings"] / srv_rounds, c["absorptions"] / srv_rounds),
              "| %.3g server rounds, %.3g flyer rounds x %.1f crossing iterations" % (srv_rounds, fly_rounds, c["dark_mirrors"] / max(fly_rounds, 1)))
    sys.exit(0)
if stage == "small":
    for cfg in (M.small(), M.small(lsepar_pola=False), M.small(n_rad=12, nz=6, n_az=8, l3D=True)):
        m = M.build_model(cfg)
        prior = run(m, 20000, None)["E_abs"]
        ref = run(m, 50000, None, prior=prior)
        for roles in (0, 1, 4, 7, 132, 148):
            r = run(m, 50000, roles, prior=prior)
            ok = r["counters"] == ref["counters"] and np.array_equal(r["sed"][4], ref["sed"][4]) and \
                np.allclose(r["E_abs"], ref["E_abs"], rtol=1e-9, atol=1e-12 * ref["E_abs"].max())
            print(cfg.l3D, cfg.lsepar_pola, "roles", roles, "OK" if ok else "MISMATCH", r["counters"]["packets"], r["kernel_ms"])
            assert ok
else:
    cfg = M.ref41()
    if "--no-pola" in sys.argv:
        cfg.lsepar_pola = False
    m = M.build_model(cfg)
    n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 20_000_000
    base = run(m, n, None)
    print("default", base["kernel_ms"])
    for em in (1, 4, 8, 16, 24, 32, 48):
        roles, ks, fi, idle = 200, 2, 16, 16
        r = run(m, n, roles, ks, fi, idle, 1 << 20, em)
        print("emit_min", em, end=" ")
        print("roles", roles, "k_short", ks, "fly_iters", fi, "fly_idle", idle, "ms", r["kernel_ms"], "crossings/pk", r["counters"]["crossings"] / n,
              "escaped+killed", r["counters"]["escaped"] + r["counters"]["killed_star"])
