"""Time one full nonhydrostatic acoustic substep (DynCore) at C384L127 size with per-kernel HIP events."""
import json, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, "tests")
import numpy as np, torch
import parity_common as P, parity_dyn as D
from gfdl_atmos_cubed_sphere_amd import lib as L
from gfdl_atmos_cubed_sphere_amd.dyn_core import DynCore, DynFlags
from gfdl_atmos_cubed_sphere_amd.layout import Bounds
import parity_nh as N
nx, npz = int(sys.argv[1]) if len(sys.argv) > 1 else 384, int(sys.argv[2]) if len(sys.argv) > 2 else 127
bd = Bounds(1, nx, 1, nx); g = P.make_grid(bd, False)
st, dp0 = D.make_state(bd, npz)
ctx = L.Context(g, npz, stream=torch.cuda.current_stream().cuda_stream)
fl = DynFlags(n_split=5, ptop=N.PTOP)
dc = DynCore(ctx, fl, dp0)
dc.set_state(st["u"], st["v"], st["w"], st["delp"], st["pt"], st["delz"], st["phis"])
dc.run(10.0); ctx.sync()
dc.set_state(st["u"], st["v"], st["w"], st["delp"], st["pt"], st["delz"], st["phis"])
t0 = time.perf_counter(); dc.run(10.0); ctx.sync(); t1 = time.perf_counter()
ctx.profile(True); dc.set_state(st["u"], st["v"], st["w"], st["delp"], st["pt"], st["delz"], st["phis"]); dc.run(10.0); rep = ctx.profile_report()
out = {"ms_per_substep": (t1 - t0) / fl.n_split * 1e3, "kernels_ms_per_substep": {k: v[1] / fl.n_split for k, v in rep.items()},
       "finite": bool(np.isfinite(dc.get_state()["w"]).all())}
print(json.dumps(out))
