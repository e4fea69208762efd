#!/usr/bin/env python3
"""bench.py -- throughput of the FV3 acoustic-substep horizontal sweeps (c_sw + d_sw) on MI355X.

One "step" = one pass of the hot path over the resident synthetic state: c_sw (all levels) ->
halo refresh of uc, vc, divg_d (periodic copy on one GPU, RCCL peer exchange across GPUs) -> d_sw
(all levels).  Workload at N=1: one doubly periodic 384 x 384 x 127 tile, nonhydrostatic, fp64
(BASELINE.json config "C384L127 ... one 384^2 tile with synthetic periodic metrics"); at N>1 every
rank owns a 384 x 384 x 127 block of a (384*px) x (384*py) doubly periodic domain (weak scaling).
metric: cell-updates/s, one cell-update = one (i,j,k) cell through one c_sw+d_sw pair.

Prints ONE JSON line (see the build contract): value, roofline (HIP-event timed dominant kernel
against its algorithmic bytes), cpu_baseline (the oracle port on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK = 8.0e12  # B/s, /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s"

# Algorithmic HBM bytes per cell of every timed launch label of the pair (fp64, NH, nord > 0, d_con = 0; DESIGN.md
# section 3a) and the levels that launch covers: the marching kernels run the "plain" levels, the LDS-tile kernels the
# sponge levels whose damping branches are not in marching form (2-3 of 127 at the reference defaults).  A launch is priced
# on the cells IT processes: frac = cells(label) * bytes / duration / 8 TB/s.
ALG = {
    # reads delp,pt,u,v,w; writes delpc,ptc,wc,uc,vc,ua,va,ut,vt,divg_d
    "c_sw": (120.0, "all"),
    # DswTransportFused: reads delp,pt,w,uc,vc; read-modify-writes mfx,mfy,cx,cy; writes delp,pt,w,crx,cry,xfx,yfx
    # = 20 arrays of SURVEY 8(d)'s d_sw list (the 16 B of zeroed heat_source / diss_est it also writes are not counted)
    "d_sw_fused": (160.0, "plain"),
    # DswMomentumFused: reads u,v,uc,vc,divg_d,crx,xfx,cry,yfx; writes u,v,delpc
    "d_sw_mom_fused": (96.0, "plain_m"),
    # the sponge levels (LDS-tile kernels): Courant numbers / transports / momentum
    "d_sw_courant": (80.0, "damp"),
    "d_sw_transport": (128.0, "damp"),
    "d_sw_momentum": (104.0, "rest_m"),
}
PAIR_ALG_BYTES = 336.0       # perfectly fused c_sw+d_sw, NH
PAIR_ALG_BYTES_DCON0 = 320.0 # ... with d_con = 0 (the flags this bench runs): heat_source is neither read nor written (section 8d: "-16")
# untimed passes before the profiled pass and the W warm-up steps (FV3_BENCH_SPINUP): the clocks of an idle MI355X take a few
# hundred launches to settle
SPINUP = int(os.environ.get("FV3_BENCH_SPINUP", "0"))
DRYRUN = os.environ.get("FV3_BENCH_DRYRUN") == "1"   # see main()
DEV = "cpu" if DRYRUN else "cuda"


def level_sets(lev, sponge_march=False):
    """the level lists fv3_dsw_levels_upload forms (csrc/fv3_api.hip): which levels go to the marching kernels.  sponge_march: the
    selection d_sw activates with uniform metrics (lev_activate: the sponge levels' nord_k = 0 / nord_w = 0 forms are in the marching kernels)"""
    npz = len(lev["nord_k"])
    damp = [k for k in range(npz) if lev["damp_vt"][k] > 1e-4 or lev["damp_t"][k] > 1e-4 or
            (lev["damp_w"][k] > 1e-5 and not (sponge_march and lev["nord_w"][k] == 0))]
    rest_m = [k for k in range(npz) if not ((lev["nord_k"][k] == 1 or (sponge_march and lev["nord_k"][k] == 0)) and
                                            not lev["damp_vt"][k] > 1e-5 and not lev["d_con_k"][k] > 1e-5)]
    return {"all": npz, "damp": len(damp), "plain": npz - len(damp), "rest_m": len(rest_m), "plain_m": npz - len(rest_m)}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--nx", type=int, default=384)
    ap.add_argument("--ny", type=int, default=0, help="rows of a rank's block when they differ from --nx (e.g. --nx 512 --ny 256: the block of the 8-GPU "
                                                      "strong-scaling run of BASELINE config 4, for FV3_BENCH_LOOPBACK=1 on one GPU)")
    ap.add_argument("--npz", type=int, default=127)
    ap.add_argument("--domain", type=int, default=0,
                    help="STRONG scaling: one doubly periodic DOMAIN x DOMAIN x npz domain (BASELINE config 4: 1024) split px x py over the "
                         "ranks; `value` is then the domain's cell-updates/s and \"scaling\" is \"strong\"")
    ap.add_argument("--strong-domain", type=int, default=None,
                    help="the domain of the strong-scaling leg every default run carries beside the weak-scaling headline (default 1024; 0: no leg)")
    ap.add_argument("--strong-steps", type=int, default=10)
    ap.add_argument("--periodic6", action="store_true", help="--gpus 6 as a 3x2 doubly periodic layout instead of the six cubed-sphere faces")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--parity-columns", action="store_true",
                    help="(the default since round 4) whole-step legs with the parity (bit-comparable) column solvers")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--hord", type=int, default=10, help="hord_mt = hord_vt = hord_tm = hord_dp (reference default 10)")
    ap.add_argument("--no-model-step", action="store_true", help="skip the SYPD (whole model step) leg")
    ap.add_argument("--no-cubed", action="store_true", help="skip the cubed-sphere leg (C384L127 face pair, whole-sphere model steps)")
    ap.add_argument("--nq", type=int, default=4, help="advected tracers in the SYPD leg")
    ap.add_argument("--model-step-multi", action="store_true", help="run the SYPD leg on N > 1 GPUs too")
    ap.add_argument("--no-general", action="store_true", help="skip the second (general-metrics) measurement of the pair")
    ap.add_argument("--general-metrics", action="store_true",
                    help="FV3_MI355X_GEOM=0: read every metric row (what a cubed-sphere gridstruct needs) instead of "
                         "using the uniform-Cartesian kernels the library selects for this doubly periodic gridstruct")
    a = ap.parse_args()
    a.parity_columns = True   # the library has one mode of the column solvers (the tolerance mode of rounds 2 - 4 was removed in round 5)
    return a


def native_oracle():
    """the CPU baseline is the C port built -O3 -march=native ON the machine that times it (the libfvo.so that
    travels with the snapshot is the parity checker's -O2 generic build).  Same sources, same -ffp-contract=off."""
    import ctypes
    import subprocess
    odir = os.path.join(ROOT, "oracle")
    out = os.path.join(odir, "_native")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libfvo_native.so")
    srcs = sorted(os.path.join(odir, f) for f in os.listdir(odir) if f.endswith(".c"))
    subprocess.check_call(["gcc", "-O3", "-march=native", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-fopenmp", "-shared",
                           "-o", so] + srcs + ["-lm"], stderr=subprocess.DEVNULL)
    return ctypes.CDLL(so)


def physical_cores():
    """physical cores of the host (SMT siblings counted once), from /proc/cpuinfo; falls back to os.cpu_count()"""
    try:
        seen, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        return len(seen) or (os.cpu_count() or 1)
    except OSError:
        return os.cpu_count() or 1


def cpu_baseline(nx, seconds):
    """The oracle's C port (OpenMP over k like dyn_core.F90:436,658) timed on this host's physical cores on a bounded
    sample of the same workload: 3 warm-up + timed repetitions for about `seconds` of CPU wall, median."""
    cores = physical_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)
    os.environ.setdefault("OMP_PROC_BIND", "spread")
    try:   # the port's per-slab work arrays (the Fortran's stack arrays) come from malloc: keep them in the per-thread arenas
        import ctypes                       # instead of mmap / munmap + page faults on every slab
        libc = ctypes.CDLL("libc.so.6")
        libc.mallopt(-3, 32 << 20)          # M_MMAP_THRESHOLD: no mmap below 32 MB
        libc.mallopt(-1, (1 << 31) - 1)     # M_TRIM_THRESHOLD: never give arena memory back
    except OSError:
        pass
    import oracle_lib as O
    try:
        O._LIB = native_oracle()
        build = "gcc -O3 -march=native -ffp-contract=off -fopenmp"
    except Exception:  # noqa: BLE001   (no compiler on the box: the shipped checker build)
        build = "gcc -O2 -mfma -ffp-contract=off -fopenmp (shipped checker build)"
    import parity_common as P
    from gfdl_atmos_cubed_sphere_amd.dyn_core import DynFlags, level_coefficients
    from gfdl_atmos_cubed_sphere_amd.layout import Bounds
    from gfdl_atmos_cubed_sphere_amd.synthetic import CSW_OUT, DSW_PAR, smooth_state
    npz = max(2, min(127, cores))
    bd = Bounds(1, nx, 1, nx)
    g = P.make_grid(bd, False)
    st = smooth_state(bd, npz, noise=0.05)
    f = {k: v for k, v in st.items()}
    for n, kind in CSW_OUT + (("mfx", "FX"), ("mfy", "FY"), ("cx", "CX"), ("cy", "CY"), ("crx", "CX"),
                              ("cry", "CY"), ("xfx", "CX"), ("yfx", "CY"), ("heat_source", "CC"),
                              ("diss_est", "CC")):
        f[n] = bd.zeros(kind, npz)
    par = dict(DSW_PAR)
    par.update(nord=1, nord_v=1, nord_w=1, nord_t=1, d2_bg=0., damp_v=0., damp_w=0., damp_t=0., d_con=0.,
               hydrostatic=0, use_cond=0)
    lev = level_coefficients(npz, DynFlags())
    keep = {k: f[k].copy(order="F") for k in ("delp", "pt", "u", "v", "w")}
    t_used, times = 0.0, []
    for rep in range(60):
        for k, v in keep.items():
            f[k][...] = v  # d_sw updates in place (reference semantics); restore the inputs
        t0 = time.perf_counter()
        O.c_sw_3d(g, npz, f, nord=1, dt2=3.0, hydrostatic=False)
        O.d_sw_3d(g, npz, par, lev, f)
        dt = time.perf_counter() - t0
        t_used += dt
        if rep >= 3:
            times.append(dt)
        if (t_used > seconds and len(times) >= 2) or (len(times) >= 10 and t_used > 0.5 * seconds):
            break
    best = float(np.median(times))
    return {"value": nx * nx * npz / best, "unit": "cell-updates/s", "cores": cores, "kind": "port", "build": build,
            "sample": f"{nx}x{nx}x{npz} doubly periodic tile, c_sw+d_sw pair, 3 warm-up + median of {len(times)} reps "
                      f"({t_used:.1f} s CPU wall), OpenMP over k on {cores} threads = the host's physical cores"}


class column_mode:
    """(rounds 2 - 4: FV3_MI355X_FAST for the contexts created inside.  The tolerance mode is gone: the whole-step legs run the one mode the
    library has, the column solvers in the reference's order)"""

    def __init__(self, fast):
        self.fast = False

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def whole_step_roofline(cells, wall_s, n_substeps, k_split, nq, hydrostatic=False):
    """algorithmic HBM bytes of one dt_atmos: per acoustic substep the c_sw + d_sw pair (336 B/cell NH, 304
    hydrostatic) + the rest of the substep (~360 B/cell NH: update_dz_c/d, both Riemann solvers, p_grad_c, nh_p_grad, halos; ~120
    hydrostatic: geopk x 2, p_grad_c, one_grad_p); per remap 144 B/cell (u, v, w, pt, delp, delz, pe / pk / peln / pkz in and out)
    + 16 per tracer; per tracer_2d call 64 B/cell of shared flux rows + 16 per tracer.  frac = bytes / wall / 8 TB/s."""
    pair, rest = (304.0, 120.0) if hydrostatic else (PAIR_ALG_BYTES, 360.0)
    per_cell = n_substeps * (pair + rest) + k_split * (144.0 + 16.0 * nq) + (k_split * (64.0 + 16.0 * nq) if nq else 0.0)
    gb = cells * per_cell / 1e9
    return {"alg_bytes_per_cell_per_dt_atmos": per_cell, "alg_GB_per_dt_atmos": round(gb, 1), "GBps": round(gb / wall_s, 1),
            "frac": round(gb * 1e9 / wall_s / HBM_PEAK, 4), "bound": "hbm",
            "formula": f"{n_substeps} x ({pair:.0f} + {rest:.0f}) + {k_split} x (144 + 16 nq) + tracer_2d, nq = {nq}"}


def model_step_leg(a, torch, dist, world, rank, px, py, bd, g, stream, fast=True, keep_fields=False):
    """SYPD leg: whole nonhydrostatic model steps (fv_dynamics.F90:460-665 k_split loop: n_split acoustic substeps,
    tracer_2d, Lagrangian_to_Eulerian) on the same tile, dt_atmos=225 s, k_split=2, n_split=5 (C384 settings)."""
    from gfdl_atmos_cubed_sphere_amd import lib as L
    from gfdl_atmos_cubed_sphere_amd import synthetic as N
    from gfdl_atmos_cubed_sphere_amd.dyn_core import DynFlags
    from gfdl_atmos_cubed_sphere_amd.fv_dynamics import FvDynamics
    from gfdl_atmos_cubed_sphere_amd.layout import Bounds
    nx, npz, nq = a.nx, a.npz, a.nq
    with column_mode(fast):
        ctx = L.Context(g, npz, stream=stream.cuda_stream)
    geom_mode = ctx.geom
    st, _ = N.balanced_nh_state(Bounds(1, nx, 1, nx), npz)
    sig = np.linspace(0.0, 1.0, npz + 1) ** 1.5
    ak, bk = N.PTOP * (1.0 - sig), sig.copy()
    k_split, n_split, dt_atmos = 2, 5, 225.0
    fv = FvDynamics(ctx, DynFlags(n_split=n_split, ptop=N.PTOP), ak, bk, nq=nq, k_split=k_split, px=px, py=py,
                    rank=rank, world=world, dist=dist if world > 1 else None)
    fv.dc.set_state(st["u"], st["v"], st["w"], st["delp"], st["pt"], st["delz"], st["phis"])
    if nq:
        fv.set_tracers(np.asfortranarray(np.random.default_rng(1).uniform(0, 1, bd.shape("A", npz) + (nq,))))

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    fv.step(dt_atmos)
    fence()
    first = {n: fv.dc.d[n].download() for n in ("u", "v", "w", "delp", "pt", "delz")} if keep_fields else None
    nrep = 3
    t0 = time.perf_counter()
    for _ in range(nrep):
        fv.step(dt_atmos)
    fence()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=DEV)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    wall = el / nrep
    w = fv.dc.d["w"].download()
    ctx.profile(True)
    fv.step(dt_atmos)
    rep = ctx.profile_report()
    ctx.profile(False)
    ctx.close()
    # the column kernels against THEIR OWN algorithmic bytes (words a column kernel must read / write once per cell, DESIGN.md
    # section 3c): what VERDICT item 8 asks to see next to the pair's roofline
    col_alg = {"riem_solver3": 72.0, "riem_solver_c": 48.0, "nh_p_grad": 64.0, "p_grad_c": 56.0, "update_dz_c": 40.0}
    cells = nx * nx * npz
    col = {}
    for k_, nb in col_alg.items():
        if k_ in rep and rep[k_][0] > 0:
            ms_call = rep[k_][1] / rep[k_][0]
            col[k_] = {"ms_per_call": round(ms_call, 4), "alg_bytes_per_cell": nb, "GBps": round(cells * nb / (ms_call * 1e-3) / 1e9, 1),
                       "frac": round(cells * nb / (ms_call * 1e-3) / HBM_PEAK, 4)}
    # the vertical remap as one unit (every remap_* launch of a Lagrangian_to_Eulerian call) against ITS algorithmic bytes, 144 + 16 nq
    # per cell (whole_step_roofline); traffic: the HBM bytes rocprofv3 counted for those launches (tools/pmc_remap.sh ->
    # profiles/hbm_traffic_remap.json, reported while the sources are the ones it was measured on)
    ms_remap = sum(v[1] for k_, v in rep.items() if k_.startswith("remap_")) / k_split
    if ms_remap > 0:
        nb = 144.0 + 16.0 * nq
        e = {"ms_per_call": round(ms_remap, 4), "alg_bytes_per_cell": nb, "GBps": round(cells * nb / (ms_remap * 1e-3) / 1e9, 1),
             "frac": round(cells * nb / (ms_remap * 1e-3) / HBM_PEAK, 4), "traffic": None}
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "hbm_traffic_remap.json")) as f_:
                tr = json.load(f_)
            from gfdl_atmos_cubed_sphere_amd import lib as _lib
            key = "lds"      # the remap with the column in LDS is the default of both modes (bit-identical to the slab kernels)
            if tr.get("build_id") == _lib.build_id() and tr.get("shape") == [nx, nx, npz, nq] and key in tr:
                e["traffic"] = tr[key]
                e["traffic_over_algorithmic"] = round(tr[key] / (cells * nb), 3)
            else:
                e["traffic_note"] = f"profiles/hbm_traffic_remap.json is of build {tr.get('build_id')}, shape {tr.get('shape')}: not reported"
        except (OSError, ValueError):
            pass
        col["remap"] = e
    return {"column_solvers": "parity kernels (bit-comparable with the oracle; the remap with the column in LDS is bit-identical to the slab kernels)",
            "_first_step_fields": first, "column_kernels": col, "geometry": {0: "general metric rows", 1: "orthogonal", 2: "orthogonal + uniform"}.get(geom_mode),
            "whole_step": whole_step_roofline(cells, wall, k_split * n_split, k_split, nq),
            "kernels_sum_ms": round(sum(v[1] for v in rep.values()), 2),
            "sypd": dt_atmos / (365.0 * wall), "wall_s_per_dt_atmos": wall, "dt_atmos_s": dt_atmos, "k_split": k_split,
            "n_split": n_split, "nq": nq, "dx_m": 26000.0, "finite": bool(np.isfinite(w).all()),
            "note": f"NOT the C384 sphere: one {nx}x{nx}x{npz} doubly periodic tile per GPU ({world} tile(s)), uniform Cartesian metrics -- the "
                    "SYPD a 6-GPU one-face-per-GPU run would have with these kernels and no cube-edge exchange; the C384 L127 sphere "
                    "measured on ONE GPU is cubed_sphere.sphere_one_gpu (top level: c384_sphere_one_gpu_sypd); "
                    f"one {nx}x{nx}x{npz} tile; a C384 sphere is 6 such tiles, "
                    "so this is the SYPD of a 6-GPU one-face-per-GPU run before cube-edge exchange cost",
            "kernels_ms_per_dt_atmos": {k: round(v[1], 3) for k, v in rep.items()}}


def cubed_sphere_leg(a, torch, stream):
    """The cubed sphere itself (grid_type 0, gnomonic C384 L127): (1) the c_sw + d_sw pair on one face -- marching kernels on
    the face interior, pass kernels on the frame along the edges; (2) whole nonhydrostatic model steps of
    the Jablonowski-Williamson baroclinic wave (test_case 13) on all six faces held by this ONE GPU, halo updates by device
    gathers: the SYPD of BASELINE's metric on one MI355X."""
    from gfdl_atmos_cubed_sphere_amd import lib as L
    from gfdl_atmos_cubed_sphere_amd.cubed_dyn import CubeHaloAdapter, MultiContext
    from gfdl_atmos_cubed_sphere_amd.cubed_sphere import CubedSphere
    from gfdl_atmos_cubed_sphere_amd.dyn_core import DynFlags, level_coefficients
    from gfdl_atmos_cubed_sphere_amd.fv_dynamics import FvDynamics
    from gfdl_atmos_cubed_sphere_amd.layout import Bounds
    from gfdl_atmos_cubed_sphere_amd.synthetic import CSW_OUT, DSW_PAR, smooth_state
    from gfdl_atmos_cubed_sphere_amd.test_cases import jablonowski_williamson, set_eta
    nx, npz = a.nx, a.npz
    npx = nx + 1
    cs = CubedSphere(npx)
    gs = [cs.gridstruct(t) for t in range(6)]
    out = {"grid": f"gnomonic equidistant C{nx} L{npz}, grid_type 0"}
    # ---- (1) the pair on one face
    ctx = L.Context(gs[0], npz, stream=stream.cuda_stream)
    bd = gs[0].bd
    st = smooth_state(Bounds(1, nx, 1, nx), npz, noise=0.05)
    d = {k: ctx.from_host(v) for k, v in st.items()}
    del st
    for n, kind in tuple(CSW_OUT) + (("mfx", "FX"), ("mfy", "FY"), ("cx", "CX"), ("cy", "CY"), ("crx", "CX"), ("cry", "CY"),
                                    ("xfx", "CX"), ("yfx", "CY"), ("delp_out", "A"), ("pt_out", "A"), ("u_out", "U"), ("v_out", "V"),
                                    ("w_out", "A"), ("heat_s", "CC"), ("diss_e", "CC")):
        d[n] = ctx.zeros(kind, npz)
    ctx.dsw_levels(level_coefficients(npz, DynFlags()))
    dt = 22.5
    par = dict(DSW_PAR)
    par.update(dt=dt, hydrostatic=0, use_cond=0, hord_mt=a.hord, hord_vt=a.hord, hord_tm=a.hord, hord_dp=a.hord)

    def pair():
        ctx.c_sw(d["delpc"], d["delp"], d["ptc"], d["pt"], d["u"], d["v"], d["w"], d["uc"], d["vc"], d["ua"], d["va"], d["wc"],
                 d["ut"], d["vt"], d["divg_d"], 1, 0.5 * dt, False)
        ctx.d_sw(par, None, d["delp"], d["pt"], d["u"], d["v"], d["w"], d["uc"], d["vc"], d["ua"], d["va"], d["divg_d"], d["mfx"],
                 d["mfy"], d["cx"], d["cy"], d["crx"], d["cry"], d["xfx"], d["yfx"], None, d["delp_out"], d["pt_out"], d["u_out"],
                 d["v_out"], d["w_out"], None, None, None)     # d_con = 0: heat_s, diss_e = NULL (dyn_core.F90:798-812)
    for _ in range(5):
        pair()
    ctx.profile(True)
    pair()
    rep = ctx.profile_report()
    ctx.profile(False)
    torch.cuda.synchronize()
    nrep = 20
    t0 = time.perf_counter()
    for _ in range(nrep):
        pair()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / nrep * 1e3
    cells = nx * nx * npz
    march = sum(v[1] for k, v in rep.items() if k in ("c_sw", "d_sw_fused", "d_sw_mom_fused"))
    out["pair_one_face"] = {"ms": ms, "cell_updates_per_s": cells / (ms * 1e-3), "alg_bytes_per_cell": PAIR_ALG_BYTES,
                            "frac_wall": cells * PAIR_ALG_BYTES / (ms * 1e-3) / HBM_PEAK,
                            "marching_kernels_ms": march, "pass_kernels_ms": sum(v[1] for v in rep.values()) - march,
                            "streams": ("ms: wall time with the frame / sponge-level passes on the side stream beside the marching kernels (two lanes, "
                                        "DESIGN 3f; FV3_MI355X_SIDE_STREAM=0: one lane); launches: the profiled pair, which runs one lane "
                                        "(per-launch events), so their sum exceeds ms"),
                            "launches": {k: [v[0], round(v[1], 4)] for k, v in rep.items()}}
    ctx.close()
    del d
    # ---- (2) whole model steps on the sphere: BASELINE configs[2] at the bench size, then configs[1] (C96 L79 hydrostatic)
    out["sphere_one_gpu"], out["sphere_one_gpu_kernels_ms_per_dt_atmos"] = sphere_steps(
        torch, stream, cs, gs, nx, npz, hydrostatic=False, k_split=2, n_split=5, dt_atmos=225.0, nrep=2, fast=not a.parity_columns)
    try:
        cs2 = CubedSphere(97)
        out["config2_c96_l79_hydrostatic"], out["config2_kernels_ms_per_dt_atmos"] = sphere_steps(torch, stream, cs2, [cs2.gridstruct(t) for t in range(6)], 96, 79,
                                                           hydrostatic=True, k_split=2, n_split=6, dt_atmos=1800.0, nrep=5)
    except Exception as e:  # noqa: BLE001
        out["config2_c96_l79_hydrostatic"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def _sum_reps(reps):
    for rep in reps:
        for v in rep.values():
            yield v[1] * 1e-3


def sphere_steps(torch, stream, cs, gs, nx, npz, hydrostatic, k_split, n_split, dt_atmos, nrep, fast=True):
    """whole fv_dynamics steps of the Jablonowski-Williamson wave on six faces held by this one GPU -> (summary, kernel ms)"""
    from gfdl_atmos_cubed_sphere_amd import lib as L
    from gfdl_atmos_cubed_sphere_amd.cubed_dyn import CubeHaloAdapter, MultiContext
    from gfdl_atmos_cubed_sphere_amd.dyn_core import DynFlags
    from gfdl_atmos_cubed_sphere_amd.fv_dynamics import FvDynamics
    from gfdl_atmos_cubed_sphere_amd.test_cases import jablonowski_williamson, set_eta
    npx = nx + 1
    cells = nx * nx * npz
    bd = gs[0].bd
    ak, bk, _, _ = set_eta(npz) if npz in (79, 127) else (None, None, None, None)
    if ak is None:
        sig = np.linspace(0.0, 1.0, npz + 1) ** 1.5
        ak, bk = 300.0 * (1.0 - sig), sig.copy()
    st = jablonowski_williamson(cs, ak, bk, hydrostatic=hydrostatic)
    cs.topo.update("A", [s_["phis"] for s_ in st])
    fl = DynFlags(n_split=n_split, hydrostatic=hydrostatic, ptop=float(ak[0]), **(dict(d_ext=0.0) if hydrostatic else {}))
    ng = bd.ng
    c = (slice(ng, ng + nx), slice(ng, ng + nx))
    for s_ in st:      # T -> theta_v (fv_dynamics.F90:323-329 hydrostatic pkz, :385-394 nonhydrostatic pkz)
        if hydrostatic:
            pe = ak[0] + np.concatenate([np.zeros(s_["delp"].shape[:2] + (1,)), np.cumsum(s_["delp"], axis=2)], axis=2)[c]
            peln = np.log(pe)
            pkz = (pe[:, :, 1:] ** fl.akap - pe[:, :, :-1] ** fl.akap) / (fl.akap * (peln[:, :, 1:] - peln[:, :, :-1]))
        else:
            pkz = ((-fl.rdgas / fl.grav) * s_["delp"][c] * s_["pt"][c] / s_["delz"]) ** fl.akap
        s_["pt"][c] = s_["pt"][c] / pkz
    # a stream per face: the launches of different faces overlap on the GPU (the column solvers of one face are 2 300
    # wavefronts, a quarter of what the chip holds), the halo gathers join and fork them (cubed_halo.CubeHalo)
    fstreams = [torch.cuda.Stream() for _ in range(6)] if os.environ.get("FV3_BENCH_FACE_STREAMS", "1") == "1" else [stream] * 6
    with column_mode(fast):
        mctx = MultiContext([L.Context(g, npz, stream=fs.cuda_stream) for g, fs in zip(gs, fstreams)])
    fv = FvDynamics(mctx, fl, ak, bk, nq=0, k_split=k_split, halo=CubeHaloAdapter(mctx, npx, topo=cs.topo))
    zero = np.zeros_like(st[0]["delp"])
    fv.dc.set_state([s_["u"] for s_ in st], [s_["v"] for s_ in st], [s_.get("w", zero) for s_ in st], [s_["delp"] for s_ in st],
                    [s_["pt"] for s_ in st], [s_.get("delz", bd.zeros("CC", npz)) for s_ in st], [s_["phis"] for s_ in st])
    del st
    fv.step(dt_atmos)
    torch.cuda.synchronize()
    # (eager launches.  Rounds 2 - 5 also replayed the step as one HIP graph: with the faces in one launch group there are six times
    # fewer, six times larger launches, the host keeps up, and the replay -- one stream, nothing overlaps -- was the slower of the two,
    # 0.061 against 0.031 s per dt_atmos at C96: removed in round 6)
    graph_note = "eager launches"
    fv.step(dt_atmos)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(nrep):
        fv.step(dt_atmos)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / nrep
    graph_note += f"; wall s per dt_atmos {wall:.4f}"
    dp = fv.dc.d["delp"].download()
    # per-kernel breakdown: ALL six faces on ONE stream, eager (on six streams the HIP-event durations of overlapping kernels add
    # up to several times the wall time and say nothing -- VERDICT r2).  Its own wall time is reported next to the sum.
    for c_ in mctx.ctxs:
        c_.set_stream(fstreams[0].cuda_stream)
    torch.cuda.synchronize()
    fv.step(dt_atmos)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fv.step(dt_atmos)
    torch.cuda.synchronize()
    wall_one = time.perf_counter() - t0
    if mctx.group:
        mctx.group.stats()
    mctx.profile(True)
    fv.step(dt_atmos)
    reps = mctx.profile_report()
    mctx.profile(False)
    grp_stats = mctx.group.stats() if mctx.group else None
    kern = {}
    for rep in reps:
        for k_, v in rep.items():
            kern[k_] = kern.get(k_, 0.0) + v[1]
    kernels = {k_: round(v, 2) for k_, v in sorted(kern.items(), key=lambda kv: -kv[1])}
    summary = {"grid": f"C{nx} L{npz}", "sypd": dt_atmos / (365.0 * wall), "wall_s_per_dt_atmos": wall, "dt_atmos_s": dt_atmos,
               "whole_step": whole_step_roofline(6 * cells, wall, k_split * fl.n_split, k_split, 0, hydrostatic),
               "kernel_breakdown": {"how": "six faces on one stream, eager launches, HIP events per launch", "wall_s": round(wall_one, 4),
                                    "kernels_sum_s": round(sum(kern_v for kern_v in _sum_reps(reps)), 4)},
               "k_split": k_split, "n_split": fl.n_split, "nq": 0, "cells": 6 * cells, "launch": graph_note,
               "column_solvers": "parity kernels",
               "face_group": ("one launch per kernel for the six faces (fv3_group); launches of the profiled step that ran all faces "
                              f"at once / alone: {grp_stats[0]} / {grp_stats[1]}") if grp_stats else "off: six launches per kernel",
               "finite": bool(all(np.isfinite(x[c]).all() for x in dp)),
               "initial_condition": "test_case 13 (Jablonowski-Williamson), " + ("hydrostatic" if hydrostatic else "nonhydrostatic"),
               "note": "all six faces on ONE MI355X (six contexts, a HIP stream per face, device-gather halo updates); one face per "
                       "GPU is the driver's multi-GPU business"}
    mctx.close()
    return summary, kernels


def cubed_six_ranks(a, torch, dist, rank, json_fd):
    """BASELINE config 5's layout: `--gpus 6`, one cubed-sphere face per rank, every cube-edge message through the library's own
    exchange (fv3_cube_halo_start / _complete: RCCL grouped send / recv over xGMI on the context's communication stream).
    Headline = the c_sw -> exchange(uc, vc, divg_d) -> d_sw pair on the six C<nx> L<npz> faces; then whole nonhydrostatic steps of the
    Jablonowski-Williamson wave (SYPD).  Never run on hardware yet (no 6-GPU node was available): the same exchange is tested in
    loopback on one GPU (tests/test_gpu_parity.py::test_cube_edge_exchange_through_rccl_loopback)."""
    import ctypes as C
    from gfdl_atmos_cubed_sphere_amd import lib as L
    from gfdl_atmos_cubed_sphere_amd.cubed_dyn import CubeNativeAdapter
    from gfdl_atmos_cubed_sphere_amd.cubed_sphere import CubedSphere
    from gfdl_atmos_cubed_sphere_amd.dyn_core import DynFlags, level_coefficients
    from gfdl_atmos_cubed_sphere_amd.fv_dynamics import FvDynamics
    from gfdl_atmos_cubed_sphere_amd.layout import Bounds
    from gfdl_atmos_cubed_sphere_amd.synthetic import CSW_OUT, DSW_PAR, smooth_state
    from gfdl_atmos_cubed_sphere_amd.test_cases import jablonowski_williamson, set_eta
    nx, npz = a.nx, a.npz
    npx = nx + 1
    cs = CubedSphere(npx)
    g = cs.gridstruct(rank)
    stream = torch.cuda.current_stream()
    with column_mode(not a.parity_columns):
        ctx = L.Context(g, npz, stream=stream.cuda_stream)
    uid = [None]
    if rank == 0:
        buf = (C.c_ubyte * 128)()
        ctx.lib.check(ctx.lib.dll.fv3_comm_get_unique_id(buf), "fv3_comm_get_unique_id")
        uid[0] = bytes(buf)
    dist.broadcast_object_list(uid, src=0)
    halo = CubeNativeAdapter(ctx, [rank], list(range(6)), rank, 6, uid[0])
    st = smooth_state(Bounds(1, nx, 1, nx), npz, noise=0.05)
    d = {k: ctx.from_host(v) for k, v in st.items()}
    del st
    for n, kind in tuple(CSW_OUT) + (("mfx", "FX"), ("mfy", "FY"), ("cx", "CX"), ("cy", "CY"), ("crx", "CX"), ("cry", "CY"),
                                    ("xfx", "CX"), ("yfx", "CY"), ("delp_out", "A"), ("pt_out", "A"), ("u_out", "U"), ("v_out", "V"),
                                    ("w_out", "A"), ("heat_s", "CC"), ("diss_e", "CC")):
        d[n] = ctx.zeros(kind, npz)
    ctx.dsw_levels(level_coefficients(npz, DynFlags()))
    dt = 22.5
    par = dict(DSW_PAR)
    par.update(dt=dt, hydrostatic=0, use_cond=0, hord_mt=a.hord, hord_vt=a.hord, hord_tm=a.hord, hord_dp=a.hord)

    def step():
        ctx.c_sw(d["delpc"], d["delp"], d["ptc"], d["pt"], d["u"], d["v"], d["w"], d["uc"], d["vc"], d["ua"], d["va"], d["wc"],
                 d["ut"], d["vt"], d["divg_d"], 1, 0.5 * dt, False)
        halo.cube.start([("C", ([d["uc"]], [d["vc"]])), ("B", [d["divg_d"]])])      # dyn_core.F90:451, :565: one message per edge
        halo.cube.finish()
        ctx.d_sw(par, None, d["delp"], d["pt"], d["u"], d["v"], d["w"], d["uc"], d["vc"], d["ua"], d["va"], d["divg_d"], d["mfx"],
                 d["mfy"], d["cx"], d["cy"], d["crx"], d["cry"], d["xfx"], d["yfx"], None, d["delp_out"], d["pt_out"], d["u_out"],
                 d["v_out"], d["w_out"], None, d["heat_s"], d["diss_e"])

    def fence():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
    for _ in range(a.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    el = time.perf_counter() - t0
    t = torch.tensor([el], dtype=torch.float64, device=DEV)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t.item())
    cells = nx * nx * npz
    finite = bool(np.isfinite(d["u_out"].download()).all())
    out = {"metric": "c_sw+d_sw cell-updates/s", "value": 6 * cells * a.steps / el, "unit": "cell-updates/s", "n_gpus": 6,
           "steps": a.steps, "warmup": a.warmup, "ms_per_step": el / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"gnomonic cubed sphere C{nx} L{npz}, one face per GPU (BASELINE config 5 layout), nonhydrostatic "
                                  f"c_sw + d_sw pair, hord {a.hord}, cube-edge exchange of uc, vc, divg_d in between",
                      "layout": "6 faces x 1x1", "halo": "fv3_cube_halo_start / _complete: RCCL grouped send / recv, one message per edge",
                      "build_id": L.build_id()},
           "finite": finite,
           "roofline": {"bound": "hbm", "kernel": "pair (whole step)", "achieved": cells * PAIR_ALG_BYTES / (el / a.steps) / 1e9,
                        "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": cells * PAIR_ALG_BYTES / (el / a.steps) / HBM_PEAK, "traffic": None,
                        "note": "per GPU: one face's algorithmic bytes over the step's wall time (exchange included)"},
           "cpu_baseline": None}
    if DRYRUN:
        out["dry_run"] = "FV3_BENCH_DRYRUN=1: host logic harness + gloo, no GPU -- plumbing only, the numbers mean nothing"
    del d
    try:        # whole model steps: BASELINE configs[2] with one face per GPU
        ak, bk, _, _ = set_eta(npz) if npz in (79, 127) else (None, None, None, None)
        if ak is None:
            sig = np.linspace(0.0, 1.0, npz + 1) ** 1.5
            ak, bk = 300.0 * (1.0 - sig), sig.copy()
        st = jablonowski_williamson(cs, ak, bk, hydrostatic=False)
        cs.topo.update("A", [s_["phis"] for s_ in st])
        s_ = st[rank]
        k_split, n_split, dt_atmos = 2, 5, 225.0
        fl = DynFlags(n_split=n_split, hydrostatic=False, ptop=float(ak[0]))
        ng = g.bd.ng
        c = (slice(ng, ng + nx), slice(ng, ng + nx))
        pkz = ((-fl.rdgas / fl.grav) * s_["delp"][c] * s_["pt"][c] / s_["delz"]) ** fl.akap
        s_["pt"][c] = s_["pt"][c] / pkz
        fv = FvDynamics(ctx, fl, ak, bk, nq=0, k_split=k_split, halo=halo, dist=dist)
        fv.dc.set_state(s_["u"], s_["v"], s_["w"], s_["delp"], s_["pt"], s_["delz"], s_["phis"])
        del st
        fv.step(dt_atmos)
        fence()
        nrep = 3
        t0 = time.perf_counter()
        for _ in range(nrep):
            fv.step(dt_atmos)
        fence()
        wall = (time.perf_counter() - t0) / nrep
        t = torch.tensor([wall], dtype=torch.float64, device=DEV)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
        out["sphere_six_gpus"] = {"grid": f"C{nx} L{npz}", "sypd": dt_atmos / (365.0 * wall), "wall_s_per_dt_atmos": wall,
                                  "dt_atmos_s": dt_atmos, "k_split": k_split, "n_split": n_split,
                                  "whole_step": whole_step_roofline(cells, wall, k_split * n_split, k_split, 0),
                                  "column_solvers": "parity kernels",
                                  "finite": bool(np.isfinite(fv.dc.d["delp"].download()[c]).all())}
    except Exception as e:  # noqa: BLE001
        out["sphere_six_gpus"] = {"error": f"{type(e).__name__}: {e}"}
    ctx.close()
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())


def main():
    # HIP maps streams onto 4 hardware queues by default; the launch stream, the sponge-level side stream and RCCL's
    # stream then share queues and the kernels meant to overlap wait for each other in queue order (measured in the
    # loopback run: 2.55 ms per step with 4 queues, 2.39 with 8).  Must be set before the HIP runtime initialises.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    a = parse()
    # the contract is ONE JSON line on stdout: native libraries (RCCL prints a version banner on fd 1 when a communicator
    # is created) must not get there, so everything but that line goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        a.gpus = world
    if DRYRUN:
        # FV3_BENCH_DRYRUN=1: the plumbing of the N-rank runs (rank layout, communicators, halo / cube-edge exchanges, the barriers, the
        # MAX over ranks, the one JSON line) WITHOUT a GPU: the logic harness of the test-suite (the kernel sources compiled for the
        # host) in place of the HIP library, gloo in place of RCCL.  What it prints is marked "dry_run" and measures nothing.
        os.environ["FV3_MI355X_SO"] = os.path.join(ROOT, "tests", "hostemu", "libfv3_hostemu.so")

        class _NoStream:
            cuda_stream = 0
        torch.cuda.set_device = lambda *x, **k: None
        torch.cuda.synchronize = lambda *x, **k: None
        torch.cuda.current_stream = lambda *x, **k: _NoStream()
        torch.cuda.Stream = lambda *x, **k: _NoStream()
    torch.cuda.set_device(local)
    # FV3_BENCH_LOOPBACK=1 (one GPU): every halo message goes through RCCL to this same rank and d_sw runs in its
    # interior / rest form -- the per-step flow of the N-GPU runs, to see what the message path costs
    loopback = world == 1 and os.environ.get("FV3_BENCH_LOOPBACK") == "1"
    if world > 1 and DRYRUN:
        dist.init_process_group("gloo")
    elif world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    elif loopback:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))

    if world == 6 and not a.periodic6:      # six ranks = the six faces of the cubed sphere (BASELINE config 5's layout)
        try:
            cubed_six_ranks(a, torch, dist, rank, json_fd)
        finally:
            dist.destroy_process_group()
        return

    from gfdl_atmos_cubed_sphere_amd import lib as L
    from gfdl_atmos_cubed_sphere_amd.dyn_core import DynFlags, level_coefficients
    from gfdl_atmos_cubed_sphere_amd.grid import doubly_periodic
    from gfdl_atmos_cubed_sphere_amd.halo import HaloExchanger, choose_layout
    from gfdl_atmos_cubed_sphere_amd.layout import Bounds
    from gfdl_atmos_cubed_sphere_amd.synthetic import CSW_OUT, DSW_PAR, smooth_state

    nx, npz = a.nx, a.npz
    px, py = choose_layout(world)
    ix, iy = rank % px, rank // px

    def block(nxb, nyb):
        """this rank's nxb x nyb block of the (nxb px) x (nyb py) doubly periodic domain (tools/fv_mp_mod.F90:276-392: one tile, layout px x py)"""
        b = Bounds(1 + ix * nxb, (ix + 1) * nxb, 1 + iy * nyb, (iy + 1) * nyb)
        return b, doubly_periodic(b, nxb * px + 1, nyb * py + 1, dx_const=26000.0, dy_const=26000.0)

    # WEAK scaling (the default, `value` of the driver's --gpus N runs): every rank holds its own nx x nx block
    # STRONG scaling (--domain D, BASELINE config 4): one D x D domain, the block shrinks with the layout
    ny = a.ny or nx
    if a.domain:
        if a.domain % px or a.domain % py:
            raise SystemExit(f"bench: --domain {a.domain} does not divide over the {px} x {py} layout")
        nx, ny = a.domain // px, a.domain // py
    bd, g = block(nx, ny)
    stream = torch.cuda.current_stream()
    cells = nx * ny * npz
    # FV3_BENCH_SPONGE=0 (diagnostic): no sponge levels, every level in the marching kernels -- NOT the headline workload
    lev = level_coefficients(npz, DynFlags(d2_bg_k1=0.0, d2_bg_k2=0.0) if os.environ.get("FV3_BENCH_SPONGE") == "0"
                             else DynFlags())
    nlev = level_sets(lev)
    build = L.build_id()
    tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    traffic_db = json.load(open(tfile)) if os.path.exists(tfile) else {}

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def setup(general, dom=None):
        """resident state + the step closure.  dom = (nx, ny, g) of another block than the headline's (the strong-scaling leg).  general: FV3_MI355X_GEOM=0, every metric row is read from memory -- what a
        cubed-sphere gridstruct needs -- instead of the uniform-Cartesian kernels the library selects for this doubly
        periodic gridstruct."""
        # the switch is read when the context uploads its gridstruct; it must not leak into the contexts created later
        # (round 2: model_step_leg ran the general-metric kernels because this was never restored)
        saved = os.environ.pop("FV3_MI355X_GEOM", None)
        if general:
            os.environ["FV3_MI355X_GEOM"] = "0"
        nxb, nyb, gb = dom if dom else (nx, ny, g)
        try:
            ctx = L.Context(gb, npz, stream=stream.cuda_stream)
        finally:
            os.environ.pop("FV3_MI355X_GEOM", None)
            if saved is not None:
                os.environ["FV3_MI355X_GEOM"] = saved
        geom = ctx.geom
        # FV3_BENCH_SPLIT=1: exercise the multi-rank flow (start / d_sw interior / finish / d_sw rest) on one GPU
        halo = HaloExchanger(ctx, px, py, rank, world, split_single=loopback or os.environ.get("FV3_BENCH_SPLIT") == "1",
                             loopback=loopback)
        st = smooth_state(Bounds(1, nxb, 1, nyb), npz, noise=0.05)  # same synthetic block on every rank
        d = {k: ctx.from_host(v) for k, v in st.items()}
        del st
        for n, kind in CSW_OUT:
            d[n] = ctx.zeros(kind, npz)
        for n, kind in (("mfx", "FX"), ("mfy", "FY"), ("cx", "CX"), ("cy", "CY"), ("crx", "CX"), ("cry", "CY"),
                        ("xfx", "CX"), ("yfx", "CY"), ("delp_out", "A"), ("pt_out", "A"), ("u_out", "U"),
                        ("v_out", "V"), ("w_out", "A"), ("heat_s", "CC"), ("diss_e", "CC")):
            d[n] = ctx.zeros(kind, npz)
        ctx.dsw_levels(lev)
        heating = bool(np.any(np.asarray(lev["d_con_k"]) > 1.0e-5))
        dt = 22.5   # C384 acoustic step: dt_atmos 225 s / k_split 2 / n_split 5
        par = dict(DSW_PAR)
        par.update(dt=dt, hydrostatic=0, use_cond=0, hord_mt=a.hord, hord_vt=a.hord, hord_tm=a.hord, hord_dp=a.hord)

        def step():
            ctx.c_sw(d["delpc"], d["delp"], d["ptc"], d["pt"], d["u"], d["v"], d["w"], d["uc"], d["vc"], d["ua"],
                     d["va"], d["wc"], d["ut"], d["vt"], d["divg_d"], 1, 0.5 * dt, False)
            dsw_args = (par, None, d["delp"], d["pt"], d["u"], d["v"], d["w"], d["uc"], d["vc"], d["ua"], d["va"],
                        d["divg_d"], d["mfx"], d["mfy"], d["cx"], d["cy"], d["crx"], d["cry"], d["xfx"], d["yfx"], None,
                        d["delp_out"], d["pt_out"], d["u_out"], d["v_out"], d["w_out"], None,
                        # d_con = 0, no do_diss_est: dyn_core reads neither (dyn_core.F90:798-812), the hosts pass NULL
                        d["heat_s"] if heating else None, d["diss_e"] if heating else None)
            # start the exchange, run the part of d_sw that reads no halo while it is in flight, complete, do the rest
            if halo.overlaps:
                pending = halo.start([(d["uc"], "V"), (d["vc"], "U"), (d["divg_d"], "B")], defer=True)
                ctx.d_sw(*dsw_args, phase="interior")
                halo.post(pending)
                halo.finish(pending)
                ctx.d_sw(*dsw_args, phase="rest")
            else:
                halo.update([(d["uc"], "V"), (d["vc"], "U"), (d["divg_d"], "B")])
                ctx.d_sw(*dsw_args)

        step.block = (nxb, nyb)
        return ctx, d, step

    def run(ctx, d, step, steps, warmup):
        """`warmup` untimed and EXACTLY `steps` timed passes of c_sw -> halo -> d_sw on the resident state, then a profiled pass (HIP
        events around every launch on its own stream) for the per-launch roofline."""
        geom = ctx.geom
        nprof = 10
        nxb, nyb = step.block
        cells = nxb * nyb * npz
        for _ in range(3 + SPINUP):  # untimed: code objects loaded, work arrays of the library allocated
            step()
        tstep = step
        for _ in range(warmup):
            tstep()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            tstep()
        fence()
        el = time.perf_counter() - t0
        # the profiled pass (HIP events around every launch, on the launch stream) AFTER the timed region: the same warm GPU -- in
        # round 4 it ran first, on the idle clocks of a fresh process, and priced the kernels 15-20 % slower than the timed steps ran
        ctx.profile(True)
        for _ in range(2):
            step()
        ctx.profile_report()
        for _ in range(nprof):
            step()
        rep = ctx.profile_report()
        ctx.profile(False)
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=DEV)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        finite = bool(np.isfinite(d["u_out"].download()).all())
        ctx.close()
        # per-launch timing: in the profiled pass the sponge-level chain runs on the main stream instead of overlapping
        # the marching kernels on the side stream, so the sum of the launches can exceed the wall time of a step
        per_launch, t_sum = {}, 0.0
        nlev = level_sets(lev, sponge_march=(geom == 2 and os.environ.get("FV3_MI355X_SPONGE_MARCH", "1") != "0"))
        for name, (n, ms) in rep.items():
            per_step = ms / nprof
            t_sum += per_step
            e = {"launches_per_step": n / nprof, "ms_per_step": per_step}
            if name in ALG and nlev[ALG[name][1]] > 0 and per_step > 0.0:
                nb, which = ALG[name]
                c = nxb * nyb * nlev[which]
                e.update(alg_bytes_per_cell=nb, levels=nlev[which], GBps=c * nb / (per_step * 1e-3) / 1e9,
                         frac=c * nb / (per_step * 1e-3) / HBM_PEAK)
            per_launch[name] = e
        priced = [k for k, v in per_launch.items() if "frac" in v]
        dom = max(priced, key=lambda k: per_launch[k]["ms_per_step"]) if priced else None
        roof = None
        if dom:
            e = per_launch[dom]
            key = f"{dom}@geom{geom}"
            tr = traffic_db.get(key) if traffic_db.get("build_id") == build else None
            roof = {"bound": "hbm", "kernel": dom, "achieved": e["GBps"], "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                    "frac": e["frac"], "traffic": tr,
                    "traffic_note": ("rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, separate passes, of this build "
                                     f"(tools/pmc_hbm_pair.sh, build {build})") if tr else
                                    (f"profiles/hbm_traffic.json is of build {traffic_db.get('build_id')}, this is {build}: "
                                     "not reported"),
                    "alg_bytes_per_cell": e["alg_bytes_per_cell"], "levels": e["levels"], "avg_ms": e["ms_per_step"],
                    "per_launch": per_launch,
                    "pair": {"alg_bytes_per_cell": PAIR_ALG_BYTES, "launches_ms": t_sum,
                             "frac_launches": cells * PAIR_ALG_BYTES / (t_sum * 1e-3) / HBM_PEAK if t_sum > 0 else None,
                             # against the wall clock of the timed region (there the sponge-level tile kernels overlap the
                             # marching kernels on a side stream, so it can beat the sum of the launches)
                             "wall_ms": el / steps * 1e3,
                             "frac_wall": cells * PAIR_ALG_BYTES / (el / steps) / HBM_PEAK,
                             # the same wall time priced at the bytes of the flags this run has (d_con = 0: no heat_source traffic)
                             "alg_bytes_per_cell_d_con_0": PAIR_ALG_BYTES_DCON0,
                             "frac_wall_d_con_0": cells * PAIR_ALG_BYTES_DCON0 / (el / steps) / HBM_PEAK}}
        return {"el": el, "value": cells * world * steps / el, "finite": finite, "roof": roof, "geom": geom}

    GEOM = {0: "general metric rows", 1: "orthogonal (angle terms not read)", 2: "orthogonal + uniform (metric terms as scalars)"}
    # all host-side setup first (state generation and uploads take seconds), then the GPU work back to back: the
    # cubed-sphere-representative pair (every metric row read, geometry mode 0), then the headline measurement
    main_set = setup(a.general_metrics)
    gm = None
    if not a.general_metrics and not a.no_general:
        try:
            gen_set = setup(True)
            gsteps = max(10, a.steps // 2)
            g0 = run(*gen_set, gsteps, max(3, a.warmup // 2))
            gm = {"gridstruct": GEOM[g0["geom"]], "value": g0["value"], "steps": gsteps, "ms_per_step": g0["el"] / gsteps * 1e3,
                  "finite": g0["finite"], "roofline": g0["roof"]}
            del gen_set
        except Exception as e:  # noqa: BLE001
            gm = {"error": f"{type(e).__name__}: {e}"}
    m = run(*main_set, a.steps, a.warmup)
    del main_set
    el, value, finite, roof, geom = m["el"], m["value"], m["finite"], m["roof"], m["geom"]

    out = {"metric": "c_sw+d_sw cell-updates/s", "value": value, "unit": "cell-updates/s", "n_gpus": world,
           "steps": a.steps, "warmup": a.warmup, "ms_per_step": el / a.steps * 1e3, "higher_is_better": True,
           "scaling": "strong" if a.domain else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": (f"ONE doubly periodic {a.domain}x{a.domain}x{npz} domain (BASELINE config 4) split {px}x{py}: {nx}x{ny} per GPU, "
                                   if a.domain else f"doubly periodic {nx}x{ny}x{npz} tile per GPU" + (" (C384L127-sized), " if (nx, ny, npz) == (384, 384, 127) else ", ")) +
                                  f"nonhydrostatic, c_sw+d_sw pair, hord {a.hord}/{a.hord}/{a.hord}/{a.hord}, nord=1, d4_bg=0.16",
                      "layout": f"{px}x{py}", "halo": ("RCCL send/recv (loopback)" if loopback else "periodic copy") if world == 1 else "RCCL send/recv",
                      # what fv3_grid_upload found in the metric arrays (fv3_grid_geom)
                      "gridstruct": GEOM[geom], "build_id": build,
                      "launch": "eager launches",
                      "sponge_levels": "off (FV3_BENCH_SPONGE=0, diagnostic)" if os.environ.get("FV3_BENCH_SPONGE") == "0"
                      else ("d2_bg_k1 0.20, d2_bg_k2 0.015 (2 levels), inside the marching kernels (uniform metrics)" if geom == 2 and
                            os.environ.get("FV3_MI355X_SPONGE_MARCH", "1") != "0" else "d2_bg_k1 0.20, d2_bg_k2 0.015 (2 levels), LDS-tile kernels"),
                      "heat_source": "d_con = 0: heat_s / diss_e = NULL (nobody reads them, dyn_core.F90:798-812); pair priced at 336 B and at 320 B"},
           "finite": finite, "roofline": roof, "general_metrics": gm}
    # BASELINE config 4 beside the weak-scaling headline: ONE strong_domain^2 x npz doubly periodic domain split over the same px x py
    # layout (1024 x 1024, 512 x 1024, 512 x 512, 512 x 256 blocks at 1, 2, 4, 8 ranks).  Every rank takes part (collective halos);
    # a failure is reported, not raised -- but it must fail on all ranks alike, so the checks are on values every rank shares.
    out["strong_scaling"] = None
    sd = a.strong_domain if a.strong_domain is not None else (0 if DRYRUN else 1024)   # (the dry run's host harness takes seconds per 16^2 step)
    if not a.domain and sd and not a.general_metrics:
        if sd % px or sd % py:
            out["strong_scaling"] = {"error": f"{sd} does not divide over the {px} x {py} layout"}
        else:
            try:
                nxs, nys = sd // px, sd // py
                _, gs_ = block(nxs, nys)
                s_set = setup(False, (nxs, nys, gs_))
                sm = run(*s_set, a.strong_steps, max(2, a.strong_steps // 3))
                del s_set
                dom_cells = sd * sd * npz
                out["strong_scaling"] = {
                    "workload": f"ONE doubly periodic {sd}x{sd}x{npz} domain (BASELINE config 4) split {px}x{py}: {nxs}x{nys} per GPU",
                    "scaling": "strong", "n_gpus": world, "steps": a.strong_steps, "ms_per_step": sm["el"] / a.strong_steps * 1e3,
                    "value": dom_cells * a.strong_steps / sm["el"], "unit": "cell-updates/s", "finite": sm["finite"],
                    "frac_wall_per_gpu": (dom_cells / world) * PAIR_ALG_BYTES / (sm["el"] / a.strong_steps) / HBM_PEAK,
                    "note": "speed-up 1 -> N = value(N) / value(1) of this object across the driver's --gpus 1, 2, 4, 8 lines "
                            "(BASELINE.md 2: >= 6x at 8); `value` of the line itself is the WEAK-scaling number"}
            except Exception as e:  # noqa: BLE001
                out["strong_scaling"] = {"error": f"{type(e).__name__}: {e}"}
    if DRYRUN:
        out["dry_run"] = "FV3_BENCH_DRYRUN=1: host logic harness + gloo, no GPU -- plumbing only, the numbers mean nothing"
    # the secondary legs must never cost the headline line: a failure there is reported, not raised
    out["model_step"] = None
    # N > 1: the SYPD leg is off unless asked for (a failure on one rank would leave the others in a collective)
    if not a.no_model_step and (world == 1 or a.model_step_multi):
        try:
            # the SYPD is quoted with the library's one mode of the column kernels (north_star's 1e-12 on whole steps holds by
            # construction: bit-comparable with the oracle)
            out["model_step"] = model_step_leg(a, torch, dist, world, rank, px, py, bd, g, stream, fast=False, keep_fields=False)
            out["model_step"].pop("_first_step_fields", None)
        except Exception as e:  # noqa: BLE001
            out["model_step"] = {"error": f"{type(e).__name__}: {e}"}
    out["cubed_sphere"] = None
    if world == 1 and not a.no_cubed:
        try:
            out["cubed_sphere"] = cubed_sphere_leg(a, torch, stream)
        except Exception as e:  # noqa: BLE001
            out["cubed_sphere"] = {"error": f"{type(e).__name__}: {e}"}
    # the numbers of the real C384 geometry beside the headline
    cs = out.get("cubed_sphere") or {}
    if isinstance(cs.get("pair_one_face"), dict):
        out["pair_one_gnomonic_c384_face_ms"] = cs["pair_one_face"].get("ms")
    if isinstance(cs.get("sphere_one_gpu"), dict):
        out["c384_sphere_one_gpu_sypd"] = cs["sphere_one_gpu"].get("sypd")
    if rank == 0 and world == 1 and not a.no_cpu:
        try:
            out["cpu_baseline"] = cpu_baseline(nx, a.cpu_seconds)
        except Exception as e:  # noqa: BLE001
            out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1 or loopback:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
