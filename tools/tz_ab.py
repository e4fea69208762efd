"""A / B of one tracer_2d sub-cycle (nq tracers) and of update_dz_d on ONE box, in ONE process, on ONE set of arrays: contexts of several
builds (FV3_AB_SO = a.so:b.so) take turns; ms per kernel label per call.  usage: FV3_AB_SO=... python tools/tz_ab.py [rounds] [nq]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from gfdl_atmos_cubed_sphere_amd import lib as L
from gfdl_atmos_cubed_sphere_amd.dyn_core import DynFlags, level_coefficients
from gfdl_atmos_cubed_sphere_amd.grid import doubly_periodic
from gfdl_atmos_cubed_sphere_amd.layout import Bounds
L.EXPORTS = ["fv3_last_error", "fv3_create"]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 4
nx, npz = int(os.environ.get("NX", 384)), int(os.environ.get("NPZ", 127))
bd = Bounds(1, nx, 1, nx)
g = doubly_periodic(bd, nx + 1, nx + 1, dx_const=26000.0, dy_const=26000.0)
sos = os.environ["FV3_AB_SO"].split(":")
libs = [L.Fv3Lib(so) for so in sos]
ctxs = [L.Context(g, npz, lib=lb, stream=torch.cuda.current_stream().cuda_stream) for lb in libs]
c0 = ctxs[0]
rng = np.random.default_rng(3)
area = 26000.0 ** 2
def fld(kind, nlev, lo, hi, extra=()):
    return c0.from_host(np.asfortranarray(rng.uniform(lo, hi, bd.shape(kind, nlev) + extra)))
d = dict(q=fld("A", npz, 0, 1, (nq,)), q_out=fld("A", npz, 0, 1, (nq,)), dp1=fld("A", npz, 900, 1100), dp1_out=fld("A", npz, 900, 1100),
         mfx=fld("FX", npz, -0.05 * 1000 * area, 0.05 * 1000 * area), mfy=fld("FY", npz, -0.05 * 1000 * area, 0.05 * 1000 * area),
         cx=fld("CX", npz, -0.3, 0.3), cy=fld("CY", npz, -0.3, 0.3), xfx=fld("CX", npz, -0.3 * area, 0.3 * area), yfx=fld("CY", npz, -0.3 * area, 0.3 * area),
         zs=fld("A", None, 0, 100), zh=fld("A", npz + 1, 100, 3e4), zh_out=fld("A", npz + 1, 100, 3e4), ws=c0.zeros("CC"),
         crx=fld("CX", npz + 1, -0.3, 0.3), cry=fld("CY", npz + 1, -0.3, 0.3), xfz=fld("CX", npz + 1, -0.3 * area, 0.3 * area),
         yfz=fld("CY", npz + 1, -0.3 * area, 0.3 * area))
sig = np.linspace(0.0, 1.0, npz + 1) ** 1.5
dp_ref = (300.0 * (1.0 - sig[1:]) - 300.0 * (1.0 - sig[:-1])) + (sig[1:] - sig[:-1]) * 1.0e5
for ctx in ctxs:
    ctx.dsw_levels(level_coefficients(npz, DynFlags()))
    ctx.set_dp_ref(dp_ref)
tot = [{} for _ in ctxs]
ks = np.ones(npz, dtype=np.int32)
for r in range(rounds + 1):
    for n, ctx in enumerate(ctxs):
        if r > 0:
            ctx.profile(True)
        for _ in range(5):
            ctx.tracer_2d_step(1, 1, ks, nq, 8, 0, 0.0, d["q"], d["q_out"], d["dp1"], d["dp1_out"], d["mfx"], d["mfy"], d["cx"], d["cy"], d["xfx"], d["yfx"])
            ctx.update_dz_d(10, d["zs"], d["zh"], d["zh_out"], d["crx"], d["cry"], d["xfz"], d["yfz"], d["ws"], 1.0 / 22.5)
        ctx.sync()
        if r > 0:
            for k, (cnt, ms) in ctx.profile_report().items():
                tot[n].setdefault(k, []).append(ms / cnt)
            ctx.profile(False)
for so, t in zip(sos, tot):
    print(os.path.basename(so), f"nq={nq}", {k: (round(float(np.median(v)), 4), round(float(np.min(v)), 4)) for k, v in t.items()})
