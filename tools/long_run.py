"""Integrate the nonhydrostatic model step (k_split x [n_split substeps, tracer_2d, remap]) for many dt_atmos on one
GPU and report stability indicators: finiteness, extrema, total air mass and tracer mass drift.
usage: long_run.py [nsteps] [nx] [npz] [nq]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import parity_dyn as D, parity_nh as N
from gfdl_atmos_cubed_sphere_amd import lib as L
from gfdl_atmos_cubed_sphere_amd.dyn_core import DynFlags
from gfdl_atmos_cubed_sphere_amd.fv_dynamics import FvDynamics
from gfdl_atmos_cubed_sphere_amd.grid import doubly_periodic
from gfdl_atmos_cubed_sphere_amd.layout import Bounds
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nx = int(sys.argv[2]) if len(sys.argv) > 2 else 384
npz = int(sys.argv[3]) if len(sys.argv) > 3 else 127
nq = int(sys.argv[4]) if len(sys.argv) > 4 else 2
dx = 26000.0
bd = Bounds(1, nx, 1, nx)
g = doubly_periodic(bd, nx + 1, nx + 1, dx_const=dx, dy_const=dx)
st, _ = D.make_state(bd, npz)
sig = np.linspace(0.0, 1.0, npz + 1) ** 1.5
ak, bk = N.PTOP * (1.0 - sig), sig.copy()
ctx = L.Context(g, npz, stream=torch.cuda.current_stream().cuda_stream)
fv = FvDynamics(ctx, DynFlags(n_split=5, ptop=N.PTOP), ak, bk, nq=nq, k_split=2)
fv.dc.set_state(st["u"], st["v"], st["w"], st["delp"], st["pt"], st["delz"], st["phis"])
r = (bd.is_, bd.ie, bd.js, bd.je)
if nq:
    q0 = np.asfortranarray(np.random.default_rng(1).uniform(0, 1, bd.shape("A", npz) + (nq,)))
    fv.set_tracers(q0)
def masses():
    dp = bd.view(fv.dc.d["delp"].download(), "A", *r)
    out = {"air": float(np.sum(dp))}
    if nq:
        q = fv.dc.d["q"].download()
        for iq in range(nq):
            out[f"q{iq}"] = float(np.sum(bd.view(q[:, :, :, iq], "A", *r) * dp))
    return out
m0 = masses()
t0 = time.perf_counter()
for n in range(nsteps):
    fv.step(225.0)
ctx.sync()
wall = time.perf_counter() - t0
s = fv.dc.get_state()
m1 = masses()
out = {"nsteps": nsteps, "simulated_hours": nsteps * 225.0 / 3600.0, "wall_s": wall, "sypd": 225.0 * nsteps / (365.0 * wall),
       "finite": bool(all(np.isfinite(s[k]).all() for k in ("u", "v", "w", "delp", "pt", "delz"))),
       "max_abs_w": float(np.max(np.abs(s["w"]))), "max_abs_u_dx": float(np.max(np.abs(s["u"]))),
       "min_delp": float(np.min(bd.view(s["delp"], "A", *r))), "max_delz": float(np.max(s["delz"])),
       "mass_rel_drift": {k: (m1[k] - m0[k]) / m0[k] for k in m0}}
print(json.dumps(out))
