"""tracer_2d sub-cycle kernel: time per dt_atmos for 1..4 tracers per wavefront (FV3_MI355X_TRACER_NT) and several nq.
Run on the GPU box: python tools/bench_tracer.py [nx npz]"""
import os
import sys

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

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 384
npz = int(sys.argv[2]) if len(sys.argv) > 2 else 127
bd = Bounds(1, nx, 1, nx)
g = doubly_periodic(bd, nx + 1, nx + 1, dx_const=26000.0, dy_const=26000.0)
st, _ = D.make_state(bd, npz)
sig = np.linspace(0.0, 1.0, npz + 1) ** 1.5
ak, bk = N.PTOP * (1.0 - sig), sig.copy()
for nq in (4, 12, 33):
    q = np.asfortranarray(np.random.default_rng(1).uniform(0, 1, bd.shape("A", npz) + (nq,)))
    for nt in (1, 2, 3, 4):
        os.environ["FV3_MI355X_TRACER_NT"] = str(nt)
        ctx = L.Context(g, npz)
        fv = FvDynamics(ctx, DynFlags(n_split=5, ptop=N.PTOP), ak, bk, nq=nq, k_split=2)
        fv.dc.set_state(st["u"], st["v"], st["w"], st["delp"], st["pt"], st["delz"], st["phis"])
        fv.set_tracers(q)
        fv.step(225.0)
        torch.cuda.synchronize()
        ctx.profile(True)
        fv.step(225.0)
        rep = ctx.profile_report()
        ctx.profile(False)
        n, ms = rep["tracer_step"][0], rep["tracer_step"][1]
        cells = nx * nx * npz
        print(f"nq={nq:3d} nt={nt}: tracer_step {n} launches {ms:8.3f} ms per dt_atmos "
              f"({ms / n:6.3f} ms each, {cells * nq / (ms / n * 1e-3) / 1e9:6.2f} G tracer-cells/s)", flush=True)
        ctx.close()
        del fv, ctx
