"""Run the c_sw -> halo -> d_sw pair a few times on one GPU (for rocprofv3 counter passes: few dispatches).
usage: run_pair.py [reps] [nx] [npz]"""
import os, sys
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
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nx = int(sys.argv[2]) if len(sys.argv) > 2 else 384
npz = int(sys.argv[3]) if len(sys.argv) > 3 else 127
bd = Bounds(1, nx, 1, nx)
g = doubly_periodic(bd, nx + 1, nx + 1, dx_const=26000.0, dy_const=26000.0)
ctx = L.Context(g, npz, stream=torch.cuda.current_stream().cuda_stream)
halo = HaloExchanger(ctx, 1, 1, 0, 1)
d = {k: ctx.from_host(v) for k, v in smooth_state(bd, npz, noise=0.05).items()}
for n, kind in P.CSW_OUT:
    d[n] = ctx.zeros(kind, npz)
for n, kind in (("mfx", "FX"), ("mfy", "FY"), ("cx", "CX"), ("cy", "CY"), ("crx", "CX"), ("cry", "CY"), ("xfx", "CX"),
                ("yfx", "CY"), ("delp_out", "A"), ("pt_out", "A"), ("u_out", "U"), ("v_out", "V"), ("w_out", "A"),
                ("heat_s", "CC"), ("diss_e", "CC")):
    d[n] = ctx.zeros(kind, npz)
ctx.dsw_levels(level_coefficients(npz, DynFlags()))
dt = 22.5
par = dict(P.DSW_PAR); par.update(dt=dt, hydrostatic=0, use_cond=0)
for _ in range(reps):
    ctx.c_sw(d["delpc"], d["delp"], d["ptc"], d["pt"], d["u"], d["v"], d["w"], d["uc"], d["vc"], d["ua"], d["va"],
             d["wc"], d["ut"], d["vt"], d["divg_d"], 1, 0.5 * dt, False)
    halo.update([(d["uc"], "V"), (d["vc"], "U"), (d["divg_d"], "B")])
    ctx.d_sw(par, None, d["delp"], d["pt"], d["u"], d["v"], d["w"], d["uc"], d["vc"], d["ua"], d["va"], d["divg_d"],
             d["mfx"], d["mfy"], d["cx"], d["cy"], d["crx"], d["cry"], d["xfx"], d["yfx"], None, d["delp_out"],
             d["pt_out"], d["u_out"], d["v_out"], d["w_out"], None, None, None)   # d_con = 0: heat_s, diss_e = NULL
ctx.sync()
print("ok")
