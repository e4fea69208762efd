"""A / B of nh_p_grad (a2b_ord4 of four fields + the gradient) on ONE box, in ONE process, on ONE set of arrays: contexts of several builds
(FV3_AB_SO = a.so:b.so) take turns; ms per kernel label per call from the library's own HIP events.  usage: FV3_AB_SO=... python tools/a2b_ab.py [rounds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from gfdl_atmos_cubed_sphere_amd import lib as L
from gfdl_atmos_cubed_sphere_amd.grid import doubly_periodic
from gfdl_atmos_cubed_sphere_amd.layout import Bounds
L.EXPORTS = ["fv3_last_error", "fv3_create"]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
nx, npz = int(os.environ.get("NX", 384)), int(os.environ.get("NPZ", 127))
bd = Bounds(1, nx, 1, nx)
g = doubly_periodic(bd, nx + 1, nx + 1, dx_const=26000.0, dy_const=26000.0)
sos = os.environ["FV3_AB_SO"].split(":")
libs = [L.Fv3Lib(so) for so in sos]
ctxs = [L.Context(g, npz, lib=lb, stream=torch.cuda.current_stream().cuda_stream) for lb in libs]
c0 = ctxs[0]
rng = np.random.default_rng(3)
def fld(kind, nlev, lo, hi):
    return c0.from_host(np.asfortranarray(rng.uniform(lo, hi, bd.shape(kind, nlev))))
d = dict(u=fld("U", npz, -20, 20), v=fld("V", npz, -20, 20), pp=fld("A", npz + 1, -50, 50), gz=fld("A", npz + 1, 100, 3e4),
         delp=fld("A", npz, 500, 1500), pk=fld("A", npz + 1, 10, 50), uc=fld("V", npz, -20, 20), vc=fld("U", npz, -20, 20))
tot = [{} for _ in ctxs]
for r in range(rounds + 1):
    for n, ctx in enumerate(ctxs):
        if r > 0:
            ctx.profile(True)
        for _ in range(10):
            ctx.nh_p_grad(d["u"], d["v"], d["pp"], d["gz"], d["delp"], d["pk"], 22.5, 0.0, gz_scale=9.80665)
            ctx.p_grad_c(11.25, d["delp"], d["pp"], d["gz"], d["uc"], d["vc"], False)
        ctx.sync()
        if r > 0:
            for k, (cnt, ms) in ctx.profile_report().items():
                tot[n].setdefault(k, []).append(ms / cnt)
            ctx.profile(False)
for so, t in zip(sos, tot):
    print(os.path.basename(so), {k: (round(float(np.median(v)), 4), round(float(np.min(v)), 4)) for k, v in t.items()})
