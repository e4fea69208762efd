"""where does a JW run with given DynFlags go wrong first: after every dt_atmos the largest |u| with its (face, i, j, k)"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from gfdl_atmos_cubed_sphere_amd import lib as L
from gfdl_atmos_cubed_sphere_amd.cubed_dyn import CubeHaloAdapter, MultiContext
from gfdl_atmos_cubed_sphere_amd.cubed_sphere import CubedSphere
from gfdl_atmos_cubed_sphere_amd.dyn_core import DynFlags
from gfdl_atmos_cubed_sphere_amd.fv_dynamics import FvDynamics
from gfdl_atmos_cubed_sphere_amd.test_cases import jablonowski_williamson, set_eta
flags = json.loads(sys.argv[1]); nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
nx = int(sys.argv[3]) if len(sys.argv) > 3 else 48; dta = float(sys.argv[4]) if len(sys.argv) > 4 else 3600.0; npz = 79; npx = nx + 1
cs = CubedSphere(npx); gs = [cs.gridstruct(t) for t in range(6)]; bd = gs[0].bd
ak, bk, _, _ = set_eta(npz)
st = jablonowski_williamson(cs, ak, bk, hydrostatic=True); cs.topo.update("A", [s_["phis"] for s_ in st])
fl = DynFlags(n_split=6, hydrostatic=True, ptop=float(ak[0]), d_ext=0.0, **flags)
ng = bd.ng; c = (slice(ng, ng + nx), slice(ng, ng + nx))
for s_ in st:
    pe = ak[0] + np.concatenate([np.zeros(s_["delp"].shape[:2] + (1,)), np.cumsum(s_["delp"], axis=2)], axis=2)[c]
    peln = np.log(pe); pkz = (pe[:, :, 1:] ** fl.akap - pe[:, :, :-1] ** fl.akap) / (fl.akap * (peln[:, :, 1:] - peln[:, :, :-1]))
    s_["pt"][c] = s_["pt"][c] / pkz
mctx = MultiContext([L.Context(g, npz) for g in gs])
fv = FvDynamics(mctx, fl, ak, bk, nq=0, k_split=2, halo=CubeHaloAdapter(mctx, npx, topo=cs.topo))
zero = np.zeros_like(st[0]["delp"])
fv.dc.set_state([s_["u"] for s_ in st], [s_["v"] for s_ in st], [zero] * 6, [s_["delp"] for s_ in st], [s_["pt"] for s_ in st],
                [bd.zeros("CC", npz)] * 6, [s_["phis"] for s_ in st])
for n in range(1, nsteps + 1):
    fv.step(dta)
    u = fv.dc.d["u"].download()
    best = (0.0, None)
    for t in range(6):
        a = np.abs(np.nan_to_num(bd.view(u[t], "U", bd.is_, bd.ie, bd.js, bd.je + 1), nan=1e99))
        m = float(a.max())
        if m > best[0]:
            i, j, k = np.unravel_index(np.argmax(a), a.shape); best = (m, (t + 1, int(i) + 1, int(j) + 1, int(k) + 1))
    print(n, "max|u| %.3e at (face, i, j, k) %s" % best, flush=True)
    if best[0] > 1e3:
        break
