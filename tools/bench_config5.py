"""BASELINE config 5 sized model step: one 768x768x79 face (doubly periodic stand-in for one C768 face), hydrostatic,
33 advected tracers, hord_tr = 8.  Prints one JSON object with the wall time per dt_atmos and the kernel breakdown.
Run on the GPU box: python tools/bench_config5.py [nx npz nq]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch

import parity_common as P
import parity_dyn as D
import parity_nh as N
from gfdl_atmos_cubed_sphere_amd import lib as L
from gfdl_atmos_cubed_sphere_amd.dyn_core import DynFlags
from gfdl_atmos_cubed_sphere_amd.fv_dynamics import FvDynamics
from gfdl_atmos_cubed_sphere_amd.grid import doubly_periodic
from gfdl_atmos_cubed_sphere_amd.layout import Bounds

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 768
npz = int(sys.argv[2]) if len(sys.argv) > 2 else 79
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 33
dt_atmos, k_split, n_split = 150.0, 2, 6
bd = Bounds(1, nx, 1, nx)
g = doubly_periodic(bd, nx + 1, nx + 1, dx_const=13000.0, dy_const=13000.0)   # C768: 13 km
st, _ = D.make_state(bd, npz)
sig = np.linspace(0.0, 1.0, npz + 1) ** 1.5
ak, bk = N.PTOP * (1.0 - sig), sig.copy()
ctx = L.Context(g, npz)
fv = FvDynamics(ctx, DynFlags(n_split=n_split, ptop=N.PTOP, hydrostatic=True), ak, bk, nq=nq, k_split=k_split)
fv.dc.set_state(st["u"], st["v"], st["w"], st["delp"], st["pt"], st["delz"], st["phis"])
qd = torch.as_tensor(fv.dc.d["q"], device="cuda")          # tracers: seeded uniform(0, 1), filled on the device
gen = torch.Generator(device="cuda")
gen.manual_seed(20260928)
qd.copy_(torch.rand(qd.shape, generator=gen, device="cuda", dtype=torch.float64))
fv.step(dt_atmos)
torch.cuda.synchronize()
nrep = 3
t0 = time.perf_counter()
for _ in range(nrep):
    fv.step(dt_atmos)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / nrep
ctx.profile(True)
fv.step(dt_atmos)
rep = ctx.profile_report()
ctx.profile(False)
delp = fv.dc.d["delp"].download()
out = {"workload": f"{nx}x{nx}x{npz} doubly periodic tile (one C768L79-face-sized block), hydrostatic, {nq} tracers, "
                   f"hord_tr 8, dt_atmos {dt_atmos} s, k_split {k_split}, n_split {n_split}",
       "wall_s_per_dt_atmos": wall, "sypd_one_face_per_gpu": dt_atmos / (365.0 * wall),
       "finite": bool(np.isfinite(delp).all()),
       "kernels_ms_per_dt_atmos": {k: round(v[1], 3) for k, v in sorted(rep.items(), key=lambda kv: -kv[1][1])}}
print(json.dumps(out))
ctx.close()
