"""Time the full nonhydrostatic model step (k_split x [n_split substeps + tracer_2d + remap]) at C384L127 size
with per-kernel HIP events; prints one JSON line.  usage: tools_substep_timing.py [nx] [npz] [nq]"""
import json, sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import parity_common as P, parity_dyn as D, parity_nh as N
from gfdl_atmos_cubed_sphere_amd import lib as L
from gfdl_atmos_cubed_sphere_amd.dyn_core import DynFlags
from gfdl_atmos_cubed_sphere_amd.fv_dynamics import FvDynamics
from gfdl_atmos_cubed_sphere_amd.grid import doubly_periodic
from gfdl_atmos_cubed_sphere_amd.layout import Bounds
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 384
npz = int(sys.argv[2]) if len(sys.argv) > 2 else 127
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dx = 26000.0                     # C384-like spacing
bd = Bounds(1, nx, 1, nx); g = doubly_periodic(bd, nx + 1, nx + 1, dx_const=dx, dy_const=dx)
st, dp0 = D.make_state(bd, npz)
sig = np.linspace(0.0, 1.0, npz + 1) ** 1.5
ak, bk = N.PTOP * (1.0 - sig), sig.copy()
ctx = L.Context(g, npz, stream=torch.cuda.current_stream().cuda_stream)
fl = DynFlags(n_split=5, ptop=N.PTOP)
fv = FvDynamics(ctx, fl, ak, bk, nq=nq, k_split=2)
dc = fv.dc
def reset():
    dc.set_state(st["u"], st["v"], st["w"], st["delp"], st["pt"], st["delz"], st["phis"])
    if nq:
        fv.set_tracers(np.asfortranarray(np.random.default_rng(1).uniform(0, 1, bd.shape("A", npz) + (nq,))))
dt_atmos = 225.0
reset(); fv.step(dt_atmos); ctx.sync()
nrep = 3
t0 = time.perf_counter()
for _ in range(nrep): fv.step(dt_atmos)
ctx.sync(); t1 = time.perf_counter()
wall = (t1 - t0) / nrep
s = dc.get_state()
finite = bool(np.isfinite(s["w"]).all() and np.isfinite(s["pt"]).all())
ctx.profile(True); fv.step(dt_atmos); rep = ctx.profile_report(); ctx.profile(False)
out = {"dt_atmos": dt_atmos, "k_split": 2, "n_split": 5, "nq": nq, "wall_s_per_dt_atmos": wall,
       "sypd_one_tile_per_gpu": dt_atmos / (365.0 * wall), "kernels_ms_per_dt_atmos": {k: round(v[1], 4) for k, v in rep.items()},
       "finite": finite, "max_abs_w": float(np.max(np.abs(s["w"])))}
print(json.dumps(out))
