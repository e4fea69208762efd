"""What about the placement of the field arrays decides the time of the marching kernels?  One process, two builds (FV3_AB_SO=base:new),
several ways of placing the same arrays: (a) one fv3_malloc each, as every host does; (b) carved out of ONE big allocation, 2 MB aligned;
(c) the same with every array 4 KB x n further (different phase); (d) one allocation each again, after the big one was freed."""
import os, sys, time, statistics, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from gfdl_atmos_cubed_sphere_amd import lib as L
from gfdl_atmos_cubed_sphere_amd import synthetic as P
from gfdl_atmos_cubed_sphere_amd.dyn_core import DynFlags, level_coefficients
from gfdl_atmos_cubed_sphere_amd.synthetic import smooth_state
from gfdl_atmos_cubed_sphere_amd.grid import doubly_periodic
from gfdl_atmos_cubed_sphere_amd.halo import HaloExchanger
from gfdl_atmos_cubed_sphere_amd.layout import Bounds
L.EXPORTS = ["fv3_last_error", "fv3_create"]
nx, npz = 384, 127
bd = Bounds(1, nx, 1, nx)
g = doubly_periodic(bd, nx + 1, nx + 1, dx_const=26000.0, dy_const=26000.0)
sos = os.environ["FV3_AB_SO"].split(":")
st = smooth_state(bd, npz, noise=0.05)
libs = [L.Fv3Lib(so) for so in sos]
ctxs = [L.Context(g, npz, lib=lb, stream=torch.cuda.current_stream().cuda_stream) for lb in libs]
c0 = ctxs[0]
for ctx in ctxs:
    ctx.dsw_levels(level_coefficients(npz, DynFlags()))
halos = [HaloExchanger(ctx, 1, 1, 0, 1) for ctx in ctxs]
names = [(k, None) for k in st] + list(tuple(P.CSW_OUT) + (("mfx", "FX"), ("mfy", "FY"), ("cx", "CX"), ("cy", "CY"), ("crx", "CX"), ("cry", "CY"), ("xfx", "CX"),
         ("yfx", "CY"), ("delp_out", "A"), ("pt_out", "A"), ("u_out", "U"), ("v_out", "V"), ("w_out", "A"), ("heat_s", "CC"), ("diss_e", "CC")))


class Carved(L.DeviceArray):
    def __init__(self, ctx, shape, ptr):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in shape)
        self.nbytes = int(np.prod(self.shape)) * 8
        self.ptr = ptr

    def free(self):
        self.ptr = None


def shape_of(n, kind):
    return st[n].shape if kind is None else bd.shape(kind, npz)


def place(mode):
    d, slab = {}, None
    if mode in ("each", "each2"):
        for n, kind in names:
            a = L.DeviceArray(c0, shape_of(n, kind))
            d[n] = a
    else:
        step = {"slab": 0, "slab4k": 4096, "slab68k": 4096 * 17}[mode]
        sizes = [int(np.prod(shape_of(n, kind))) * 8 for n, kind in names]
        al = 1 << 21
        total = sum((s + al - 1) // al * al + al for s in sizes) + al
        slab = L.DeviceArray(c0, (total // 8,))
        off = (slab.ptr + al - 1) // al * al - slab.ptr
        for i, ((n, kind), s) in enumerate(zip(names, sizes)):
            d[n] = Carved(c0, shape_of(n, kind), slab.ptr + off + i * step)
            off += (s + al - 1) // al * al + al
    for n, kind in names:
        if kind is None:
            d[n].upload(st[n])
        else:
            d[n].zero()
    return d, slab


dt = 22.5
par = dict(P.DSW_PAR); par.update(dt=dt, hydrostatic=0, use_cond=0, hord_mt=10, hord_vt=10, hord_tm=10, hord_dp=10)


def pair(ctx, halo, d):
    ctx.c_sw(d["delpc"], d["delp"], d["ptc"], d["pt"], d["u"], d["v"], d["w"], d["uc"], d["vc"], d["ua"], d["va"],
             d["wc"], d["ut"], d["vt"], d["divg_d"], 1, 0.5 * dt, False)
    halo.update([(d["uc"], "V"), (d["vc"], "U"), (d["divg_d"], "B")])
    ctx.d_sw(par, None, d["delp"], d["pt"], d["u"], d["v"], d["w"], d["uc"], d["vc"], d["ua"], d["va"], d["divg_d"],
             d["mfx"], d["mfy"], d["cx"], d["cy"], d["crx"], d["cry"], d["xfx"], d["yfx"], None, d["delp_out"],
             d["pt_out"], d["u_out"], d["v_out"], d["w_out"], None, d["heat_s"], d["diss_e"])


hold = []
for mode in sys.argv[1:] or ["each", "slab", "slab4k", "slab68k", "each2"]:
    if mode.startswith("hold"):     # keep another <n> MB allocated from here on: the following placements move
        hold.append(L.DeviceArray(c0, (int(mode[4:]) * (1 << 20) // 8,)))
        c0._buffers.clear()
        continue
    d, slab = place(mode)
    out = []
    for n, (ctx, halo) in enumerate(zip(ctxs, halos)):
        for _ in range(8):
            pair(ctx, halo, d)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            pair(ctx, halo, d)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 20 * 1e3
        ctx.profile(True)
        for _ in range(5):
            pair(ctx, halo, d)
        rep = ctx.profile_report()
        ctx.profile(False)
        out.append("%s: wall %.3f c_sw %.3f fused %.3f mom %.3f" % (os.path.basename(sos[n]), wall, rep["c_sw"][1] / rep["c_sw"][0],
                                                                   rep["d_sw_fused"][1] / rep["d_sw_fused"][0], rep["d_sw_mom_fused"][1] / rep["d_sw_mom_fused"][0]))
    print("%-8s" % mode, " | ".join(out), " delp=%x cx=%x" % (d["delp"].ptr, d["cx"].ptr), flush=True)
    for a in list(d.values()):
        if not isinstance(a, Carved):
            a.free()
    if slab is not None:
        slab.free()
    c0._buffers.clear()
