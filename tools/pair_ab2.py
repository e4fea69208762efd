"""A / B of the headline pair on ONE box, in ONE process, on ONE set of arrays: contexts of several builds of the library (FV3_AB_SO = paths
separated by ':') share the device arrays of the first, and the builds take turns -- so neither the placement of the arrays nor the state
of the box differs between them.  usage: FV3_AB_SO=a.so:b.so python tools/pair_ab2.py [rounds] [pairs per turn]
PAIR_NO_HEAT=1: heat_s / diss_e = NULL (d_con = 0: nobody reads them)."""
import os, sys, time, statistics
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
L.EXPORTS = ["fv3_last_error", "fv3_create"]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nx, npz = int(os.environ.get("NX", 384)), int(os.environ.get("NPZ", 127))
bd = Bounds(1, nx, 1, nx)
g = doubly_periodic(bd, nx + 1, nx + 1, dx_const=26000.0, dy_const=26000.0)
sos = os.environ["FV3_AB_SO"].split(":")
no_heat = bool(int(os.environ.get("PAIR_NO_HEAT", "0")))
st = smooth_state(bd, npz, noise=0.05)
# "path@NAME=VALUE,NAME2=VALUE2": environment of that build's context (read by fv3_create); FV3_MI355X_ is prefixed to bare names
envs = [dict(kv.split("=") for kv in so.split("@")[1].split(",")) if "@" in so else {} for so in sos]
labels = [os.path.basename(so) for so in sos]
sos = [so.split("@")[0] for so in sos]
_cache = {}
libs = [_cache.setdefault(so, L.Fv3Lib(so)) for so in sos]
ctxs = []
for lb, env in zip(libs, envs):
    keys = [(k if k.startswith("FV3_") else "FV3_MI355X_" + k) for k in env]
    for k, v in zip(keys, env.values()):
        os.environ[k] = v
    ctxs.append(L.Context(g, npz, lib=lb, stream=torch.cuda.current_stream().cuda_stream))
    for k in keys:
        del os.environ[k]
c0 = ctxs[0]
d = {k: c0.from_host(v) for k, v in st.items()}
for n, kind in tuple(P.CSW_OUT) + (("mfx", "FX"), ("mfy", "FY"), ("cx", "CX"), ("cy", "CY"), ("crx", "CX"), ("cry", "CY"), ("xfx", "CX"),
                                 ("yfx", "CY"), ("delp_out", "A"), ("pt_out", "A"), ("u_out", "U"), ("v_out", "V"), ("w_out", "A"),
                                 ("heat_s", "CC"), ("diss_e", "CC")):
    d[n] = c0.zeros(kind, npz)
dt = 22.5
par = dict(P.DSW_PAR); par.update(dt=dt, hydrostatic=0, use_cond=0, hord_mt=10, hord_vt=10, hord_tm=10, hord_dp=10)
halos = []
for ctx in ctxs:
    ctx.dsw_levels(level_coefficients(npz, DynFlags(n_sponge=int(os.environ.get("PAIR_NSPONGE", "1")))))   # PAIR_NSPONGE=-1: no sponge levels
    halos.append(HaloExchanger(ctx, 1, 1, 0, 1))


def pair(ctx, halo):
    ctx.c_sw(d["delpc"], d["delp"], d["ptc"], d["pt"], d["u"], d["v"], d["w"], d["uc"], d["vc"], d["ua"], d["va"],
             d["wc"], d["ut"], d["vt"], d["divg_d"], 1, 0.5 * dt, False)
    halo.update([(d["uc"], "V"), (d["vc"], "U"), (d["divg_d"], "B")])
    ctx.d_sw(par, None, d["delp"], d["pt"], d["u"], d["v"], d["w"], d["uc"], d["vc"], d["ua"], d["va"], d["divg_d"],
             d["mfx"], d["mfy"], d["cx"], d["cy"], d["crx"], d["cry"], d["xfx"], d["yfx"], None, d["delp_out"],
             d["pt_out"], d["u_out"], d["v_out"], d["w_out"], None, None if no_heat else d["heat_s"], None if no_heat else d["diss_e"])


walls = [[] for _ in ctxs]
kern = [dict() for _ in ctxs]
for ctx, halo in zip(ctxs, halos):
    for _ in range(5):
        pair(ctx, halo)
torch.cuda.synchronize()
for rnd in range(rounds):
    for n, (ctx, halo) in enumerate(zip(ctxs, halos)):
        for _ in range(3):
            pair(ctx, halo)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(npairs):
            pair(ctx, halo)
        torch.cuda.synchronize()
        walls[n].append((time.perf_counter() - t0) / npairs * 1e3)
        ctx.profile(True)
        for _ in range(5):
            pair(ctx, halo)
        rep = ctx.profile_report()
        ctx.profile(False)
        for k, v in rep.items():
            kern[n].setdefault(k, []).append(v[1] / v[0])
for n, so in enumerate(sos):
    w = walls[n]
    print("%-44s wall ms: median %.4f min %.4f max %.4f | " % (labels[n], statistics.median(w), min(w), max(w)) +
          "  ".join("%s %.4f (%.4f-%.4f)" % (k.replace("d_sw_", ""), statistics.median(v), min(v), max(v)) for k, v in kern[n].items()), flush=True)
