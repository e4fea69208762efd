"""time the two Riemann solvers alone (parity kernels / fast mode) on a C384L127-sized tile: ms per call, fraction of their own
algorithmic HBM roofline (72 / 48 B per cell, bench.py column_kernels)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    from gfdl_atmos_cubed_sphere_amd import lib as L
    from gfdl_atmos_cubed_sphere_amd.grid import doubly_periodic
    from gfdl_atmos_cubed_sphere_amd.layout import Bounds
    from gfdl_atmos_cubed_sphere_amd.lib import GRAV, nh_consts
    from gfdl_atmos_cubed_sphere_amd.synthetic import PTOP, nh_state
    nx = int(os.environ.get("NX", 384)); km = int(os.environ.get("KM", 127))
    bd = Bounds(1, nx, 1, nx)
    g = doubly_periodic(bd, nx + 1, nx + 1, dx_const=26000.0, dy_const=26000.0)
    s = nh_state(bd, km)
    cn = nh_consts(PTOP)
    rng = np.random.default_rng(3)
    for mode, a_imp in (("slab kernels", 1.0), ("lds (bit-identical)", 1.0),
                        ("slab kernels, SIM a_imp 0.75", 0.75), ("lds, SIM a_imp 0.75", 0.75))[int(os.environ.get("RT_FIRST", 0)):int(os.environ.get("RT_LAST", 4))]:
        cn = nh_consts(PTOP, a_imp=a_imp)
        os.environ.pop("FV3_MI355X_RIEM_LDS", None)
        if mode.startswith("slab"):
            os.environ["FV3_MI355X_RIEM_LDS"] = "0"
        ctx = L.Context(g, km)
        os.environ.pop("FV3_MI355X_RIEM_LDS", None)
        d = dict(zs=ctx.from_host(s["zs"]), hs=ctx.from_host(np.asfortranarray(s["zs"] * GRAV)), w=ctx.from_host(s["w"]), pt=ctx.from_host(s["pt"]),
                 delp=ctx.from_host(s["delp"]), zh=ctx.from_host(s["zh"]), gz=ctx.from_host(s["zh"]), delz=ctx.zeros("CC", km),
                 ppe=ctx.zeros("A", km + 1), pk3=ctx.zeros("A", km + 1), pef=ctx.zeros("A", km + 1),
                 ws=ctx.from_host(np.asfortranarray(0.1 * rng.uniform(-1, 1, bd.shape("CC")))),
                 wsA=ctx.from_host(np.asfortranarray(0.1 * rng.uniform(-1, 1, bd.shape("A")))))
        zh0, w0, gz0 = s["zh"], s["w"], s["zh"]

        def r3():
            ctx.riem_solver3(22.5, cn, d["zs"], d["w"], d["delz"], d["pt"], d["delp"], d["zh"], None, d["ppe"], d["pk3"], None, None, d["ws"],
                             False, False, False)

        def rc():
            ctx.riem_solver_c(11.25, cn, d["hs"], d["w"], d["pt"], d["delp"], d["gz"], d["pef"], d["wsA"])
        out = {}
        for name, fn in (("riem_solver3", r3), ("riem_solver_c", rc)):
            for _ in range(3):
                d["zh"].upload(zh0); d["w"].upload(w0); d["gz"].upload(gz0)
                fn()
            torch.cuda.synchronize()
            ts = []
            for _ in range(10):
                d["zh"].upload(zh0); d["w"].upload(w0); d["gz"].upload(gz0)
                torch.cuda.synchronize()
                t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            out[name] = min(ts) * 1e3
        cells = nx * nx * km
        print(mode, {k: round(v, 4) for k, v in out.items()},
              "frac", round(cells * 72 / (out["riem_solver3"] * 1e-3) / 8e12, 3), round(cells * 48 / (out["riem_solver_c"] * 1e-3) / 8e12, 3),
              "finite", bool(np.isfinite(d["zh"].download()).all()))
        ctx.close()


if __name__ == "__main__":
    main()
