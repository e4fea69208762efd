"""BASELINE config 4's initial condition as a (dry-dynamics) simulation on one GPU: the doubly periodic supercell sounding with its warm
bubble (test_case 17), nonhydrostatic, water vapour carried as tracer 1 with its virtual effect; no microphysics in this path, so what
develops is the buoyant thermal of the bubble, not the storm.  Every `--every` s: max w, max |u|, the global air mass, the vapour mass."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=256)
    ap.add_argument("--npz", type=int, default=64)
    ap.add_argument("--dx", type=float, default=500.0)
    ap.add_argument("--minutes", type=float, default=20.0)
    ap.add_argument("--dt-atmos", type=float, default=6.0)
    ap.add_argument("--n-split", type=int, default=8)   # c_s dt / dx = 0.5 at dx = 500 m (1.0 goes unstable within 20 steps)
    ap.add_argument("--every", type=float, default=120.0)
    ap.add_argument("--flags", type=str, default="", help="JSON of DynFlags overrides")
    a = ap.parse_args()
    from gfdl_atmos_cubed_sphere_amd import lib as L
    from gfdl_atmos_cubed_sphere_amd.dyn_core import DynFlags
    from gfdl_atmos_cubed_sphere_amd.fv_dynamics import FvDynamics
    from gfdl_atmos_cubed_sphere_amd.grid import doubly_periodic
    from gfdl_atmos_cubed_sphere_amd.layout import Bounds, periodic_fill
    from gfdl_atmos_cubed_sphere_amd.test_cases import supercell
    nx, npz = a.nx, a.npz
    bd = Bounds(1, nx, 1, nx)
    g = doubly_periodic(bd, nx + 1, nx + 1, dx_const=a.dx, dy_const=a.dx)
    sig = np.linspace(0.0, 1.0, npz + 1) ** 1.2
    ptop = 5000.0                                           # ~20 km lid
    ak, bk = ptop * (1.0 - sig), sig.copy()
    st = supercell(bd, npz, ak, bk, a.dx, a.dx, dt_amp=2.0, dt_rad=10.0e3)
    q = st.pop("q")
    for n, kind in (("u", "U"), ("v", "V"), ("w", "A"), ("delp", "A"), ("pt", "A")):
        for k in range(npz):
            periodic_fill(bd, st[n][:, :, k], kind)
    for k in range(npz):
        periodic_fill(bd, q[:, :, k, 0], "A")
    fl = DynFlags(n_split=a.n_split, ptop=ptop, **(json.loads(a.flags) if a.flags else {}))
    ctx = L.Context(g, npz)
    fv = FvDynamics(ctx, fl, ak, bk, nq=1, k_split=1, adiabatic=False, c2l_ord=2)
    fv.dc.set_state(st["u"], st["v"], st["w"], st["delp"], st["pt"], st["delz"], st["phis"])
    fv.set_tracers(q)
    ng = bd.ng
    c = (slice(ng, ng + nx), slice(ng, ng + nx))
    t_env = st["pt"][ng, ng, :].copy()                      # the sounding away from the bubble

    def diag(t):
        d = fv.dc.d
        w, dp, T, qq, u = d["w"].download(), d["delp"].download(), d["pt"].download(), d["q"].download(), d["u"].download()
        return {"t_s": t, "w_max": float(w[c].max()), "w_min": float(w[c].min()), "u_max": float(np.abs(u[ng:ng + nx, ng:ng + nx + 1]).max()),
                "dT_max": float((T[c] - t_env[None, None, :]).max()), "mass": float(dp[c].sum()), "vapour_mass": float((dp[c] * qq[c][..., 0]).sum()),
                "finite": bool(np.isfinite(w[c]).all())}
    out = [diag(0.0)]
    nsteps = int(round(a.minutes * 60.0 / a.dt_atmos))
    every = max(1, int(round(a.every / a.dt_atmos)))
    t0 = time.perf_counter()
    for n in range(1, nsteps + 1):
        fv.step_from_temperature(a.dt_atmos)
        if n % every == 0:
            out.append(diag(n * a.dt_atmos))
            if not out[-1]["finite"]:
                break
    ctx.sync()
    wall = time.perf_counter() - t0
    m0, v0 = out[0]["mass"], out[0]["vapour_mass"]
    print(json.dumps({"config": f"doubly periodic {nx} x {nx} x {npz}, dx = {a.dx} m, supercell IC (test_case 17), nonhydrostatic dry dynamics + vapour tracer, "
                                f"dt_atmos {a.dt_atmos} s x n_split {a.n_split}", "minutes": a.minutes, "steps": nsteps, "wall_s": wall,
                      "mass_drift_rel": abs(out[-1]["mass"] - m0) / m0, "vapour_mass_drift_rel": abs(out[-1]["vapour_mass"] - v0) / v0,
                      "w_max_overall": max(s["w_max"] for s in out), "build_id": L.build_id(), "series": out}))
    ctx.close()


if __name__ == "__main__":
    main()
