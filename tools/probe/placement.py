"""Does the placement of the field arrays decide the time of the d_sw kernels?  The same library, fresh contexts one after the other, the
addresses of the arrays beside the per-kernel times.  usage: FV3_AB_SO=lib.so python tools/probe/placement.py [contexts]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gfdl_atmos_cubed_sphere_amd import lib as L
from gfdl_atmos_cubed_sphere_amd import synthetic as P
from gfdl_atmos_cubed_sphere_amd.dyn_core import DynFlags, level_coefficients
from gfdl_atmos_cubed_sphere_amd.synthetic import smooth_state
from gfdl_atmos_cubed_sphere_amd.grid import doubly_periodic
from gfdl_atmos_cubed_sphere_amd.halo import HaloExchanger
from gfdl_atmos_cubed_sphere_amd.layout import Bounds
L.EXPORTS = ["fv3_last_error", "fv3_create"]
nctx = int(sys.argv[1]) if len(sys.argv) > 1 else 6
nx, npz = 384, 127
bd = Bounds(1, nx, 1, nx)
g = doubly_periodic(bd, nx + 1, nx + 1, dx_const=26000.0, dy_const=26000.0)
so = os.environ.get("FV3_AB_SO", os.path.join(ROOT, "gfdl_atmos_cubed_sphere_amd", "csrc", "libfv3_mi355x.so"))
st = smooth_state(bd, npz, noise=0.05)
lib = L.Fv3Lib(so)
keep = []
for n in range(nctx):
    ctx = L.Context(g, npz, lib=lib, stream=torch.cuda.current_stream().cuda_stream)
    halo = HaloExchanger(ctx, 1, 1, 0, 1)
    d = {k: ctx.from_host(v) for k, v in st.items()}
    for nm, kind in tuple(P.CSW_OUT) + (("mfx", "FX"), ("mfy", "FY"), ("cx", "CX"), ("cy", "CY"), ("crx", "CX"), ("cry", "CY"), ("xfx", "CX"),
                                     ("yfx", "CY"), ("delp_out", "A"), ("pt_out", "A"), ("u_out", "U"), ("v_out", "V"), ("w_out", "A"),
                                     ("heat_s", "CC"), ("diss_e", "CC")):
        d[nm] = ctx.zeros(kind, npz)
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
    ctx.profile(True)
    for _ in range(10):
        pair()
    rep = ctx.profile_report()
    ctx.profile(False)
    t = {k: round(v[1] / v[0], 4) for k, v in rep.items()}
    names = ["delp", "pt", "w", "uc", "vc", "crx", "xfx", "cry", "yfx", "cx", "cy", "mfx", "mfy", "delp_out", "pt_out", "w_out", "heat_s", "diss_e", "u", "v"]
    addrs = " ".join("%s=%x" % (k, d[k].ptr) for k in names)
    print("ctx", n, "fused", t.get("d_sw_fused"), "mom", t.get("d_sw_mom_fused"), "c_sw", t.get("c_sw"), flush=True)
    print("   ", addrs, flush=True)
    if os.environ.get("KEEP"):
        keep.append((ctx, d))   # do not free: the next context gets fresh memory
    else:
        ctx.close()
