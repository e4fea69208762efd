"""A / B of the headline pair on ONE box: the same loop through two builds of the library (FV3_AB_SO = paths separated by ':'),
alternating, per-kernel HIP-event times and the wall time per pair.  usage: FV3_AB_SO=a.so:b.so python tools/pair_ab.py [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gfdl_atmos_cubed_sphere_amd import lib as L
from gfdl_atmos_cubed_sphere_amd import synthetic as P
from gfdl_atmos_cubed_sphere_amd.dyn_core import DynFlags, level_coefficients
from gfdl_atmos_cubed_sphere_amd.synthetic import smooth_state
from gfdl_atmos_cubed_sphere_amd.grid import doubly_periodic
from gfdl_atmos_cubed_sphere_amd.halo import HaloExchanger
from gfdl_atmos_cubed_sphere_amd.layout import Bounds
L.EXPORTS = ["fv3_last_error", "fv3_create"]          # older builds export fewer entry points
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
nx, npz = 384, 127
bd = Bounds(1, nx, 1, nx)
g = doubly_periodic(bd, nx + 1, nx + 1, dx_const=26000.0, dy_const=26000.0)
sos = os.environ["FV3_AB_SO"].split(":")
st = smooth_state(bd, npz, noise=0.05)
for rnd in range(2):
    for so in sos:
        lib = L.Fv3Lib(so)
        ctx = L.Context(g, npz, lib=lib, stream=torch.cuda.current_stream().cuda_stream)
        halo = HaloExchanger(ctx, 1, 1, 0, 1)
        d = {k: ctx.from_host(v) for k, v in st.items()}
        for n, kind in tuple(P.CSW_OUT) + (("mfx", "FX"), ("mfy", "FY"), ("cx", "CX"), ("cy", "CY"), ("crx", "CX"), ("cry", "CY"), ("xfx", "CX"),
                                         ("yfx", "CY"), ("delp_out", "A"), ("pt_out", "A"), ("u_out", "U"), ("v_out", "V"), ("w_out", "A"),
                                         ("heat_s", "CC"), ("diss_e", "CC")):
            d[n] = ctx.zeros(kind, npz)
        ctx.dsw_levels(level_coefficients(npz, DynFlags()))
        dt = 22.5
        par = dict(P.DSW_PAR); par.update(dt=dt, hydrostatic=0, use_cond=0, hord_mt=10, hord_vt=10, hord_tm=10, hord_dp=10)

        def pair():
            ctx.c_sw(d["delpc"], d["delp"], d["ptc"], d["pt"], d["u"], d["v"], d["w"], d["uc"], d["vc"], d["ua"], d["va"],
                     d["wc"], d["ut"], d["vt"], d["divg_d"], 1, 0.5 * dt, False)
            halo.update([(d["uc"], "V"), (d["vc"], "U"), (d["divg_d"], "B")])
            ctx.d_sw(par, None, d["delp"], d["pt"], d["u"], d["v"], d["w"], d["uc"], d["vc"], d["ua"], d["va"], d["divg_d"],
                     d["mfx"], d["mfy"], d["cx"], d["cy"], d["crx"], d["cry"], d["xfx"], d["yfx"], None, d["delp_out"],
                     d["pt_out"], d["u_out"], d["v_out"], d["w_out"], None, d["heat_s"], d["diss_e"])
        for _ in range(10):
            pair()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            pair()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps * 1e3
        ctx.profile(True)
        for _ in range(10):
            pair()
        rep = ctx.profile_report()
        ctx.profile(False)
        print(os.path.basename(so), "wall ms", round(wall, 4), {k: round(v[1] / v[0], 4) for k, v in rep.items()}, flush=True)
        ctx.close()
