"""Rows per segment of the three marching kernels of the headline pair: per-kernel times for a list of settings (one fresh context each).
usage: FV3_AB_SO=lib.so python tools/probe/tj_sweep.py "CSW=24,FUSED=55,MOM=55" "CSW=48" ..."""
import os, sys
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
nx, npz = int(os.environ.get("NX", 384)), int(os.environ.get("NPZ", 127))
bd = Bounds(1, nx, 1, nx)
g = doubly_periodic(bd, nx + 1, nx + 1, dx_const=26000.0, dy_const=26000.0)
so = os.environ.get("FV3_AB_SO", os.path.join(ROOT, "gfdl_atmos_cubed_sphere_amd", "csrc", "libfv3_mi355x.so"))
st = smooth_state(bd, npz, noise=0.05)
lib = L.Fv3Lib(so)
for setting in sys.argv[1:] or [""]:
    for k in list(os.environ):
        if k.startswith("FV3_MI355X_MARCH_TJ"):
            del os.environ[k]
    for kv in filter(None, setting.split(",")):
        k, v = kv.split("=")
        os.environ[k if k.startswith("FV3_") else "FV3_MI355X_MARCH_TJ_" + k] = v
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
    import time
    t0 = time.perf_counter()
    for _ in range(30):
        pair()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 30 * 1e3
    ctx.profile(True)
    for _ in range(10):
        pair()
    rep = ctx.profile_report()
    ctx.profile(False)
    print("%-40s wall %.4f " % (setting, wall), {k: round(v[1] / v[0], 4) for k, v in rep.items()}, flush=True)
    ctx.close()
