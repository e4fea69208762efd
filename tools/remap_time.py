"""time Lagrangian_to_Eulerian alone (slab kernels, FV3_MI355X_REMAP_LDS=0 / the column in LDS, the default) on a C384L127-sized tile with nq tracers: ms per kernel label per call
(fv3_profile events).  NX, KM, NQ from the environment."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import parity_common as P
    import parity_nh as N
    import parity_remap as R
    from gfdl_atmos_cubed_sphere_amd import lib as L
    from gfdl_atmos_cubed_sphere_amd.layout import Bounds
    from gfdl_atmos_cubed_sphere_amd.lib import CP_AIR, GRAV, KAPPA, RDGAS
    nx, km, nq = int(os.environ.get("NX", 384)), int(os.environ.get("KM", 127)), int(os.environ.get("NQ", 4))
    bd = Bounds(1, nx, 1, nx)
    g = P.make_grid(bd, False)
    f, ak, bk = R.remap_state(bd, km, nq)
    par = dict(last_step=0, hydrostatic=0, adiabatic=1, nq=nq, kord_mt=8, kord_wz=8, kord_tm=-8, sphum=1 if nq else 0, akap=KAPPA, ptop=N.PTOP,
               rdgas=RDGAS, grav=GRAV, cv_air=CP_AIR - RDGAS, r_vir=0.6077, cp=CP_AIR, t_min=184.0, kord_tr=[8] * nq)
    for lds in (False, True):
        os.environ["FV3_MI355X_REMAP_LDS"] = "2" if lds else "0"
        ctx = L.Context(g, km)
        ctx.set_ak_bk(ak, bk)
        d = {k: ctx.from_host(v) for k, v in f.items()}
        tot = {}
        for rep in range(6):
            for k, v in f.items():
                d[k].upload(v)
            ctx.sync()
            if rep >= 2:
                ctx.profile(True)
            ctx.lagrangian_to_eulerian(par, d["ps"], d["pe"], d["delp"], d["pkz"], d["pk"], d["u"], d["v"], d["w"], d["delz"], d["pt"], d.get("q"),
                                       d["peln"], d["omga"], d["ws"])
            ctx.sync()
            if rep >= 2:
                for label, (cnt, ms) in ctx.profile_report().items():
                    tot[label] = tot.get(label, 0.0) + ms / 4.0
                ctx.profile(False)
        print("lds" if lds else "slabs", {k: round(v, 4) for k, v in tot.items()}, "sum", round(sum(tot.values()), 4))
        ctx.close()


if __name__ == "__main__":
    main()
