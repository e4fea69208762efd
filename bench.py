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

# algorithmic HBM bytes per cell-update (fp64, NH, nord>0, d_con=0 defaults), DESIGN.md section 4
ALG_BYTES = {
    "c_sw": 120.0,           # reads delp,pt,u,v,w; writes delpc,ptc,wc,uc,vc,ua,va,ut,vt,divg_d
    "d_sw_courant": 80.0,    # reads uc,vc,cx,cy; writes crx,xfx,cry,yfx,cx,cy
    "d_sw_transport": 128.0, # reads delp,pt,w,crx,xfx,cry,yfx,mfx,mfy; writes delp,pt,w,mfx,mfy,heat_s,diss_e
    "d_sw_momentum": 104.0,  # reads u,v,uc,vc,divg_d,crx,xfx,cry,yfx,(delp',heat_s when d_con>0); writes u,v,delpc
}
# kernels that together do the work of one logical kernel (the marching transports are one launch per field)
GROUP = {"d_sw_delp": "d_sw_transport", "d_sw_w": "d_sw_transport", "d_sw_pt": "d_sw_transport",
         "d_sw_qcon": "d_sw_transport", "d_sw_fused": "d_sw_transport", "d_sw_ke": "d_sw_momentum", "d_sw_mom_fused": "d_sw_momentum", "d_sw_vort": "d_sw_momentum"}
PAIR_ALG_BYTES = 336.0       # perfectly fused c_sw+d_sw, NH


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--nx", type=int, default=384)
    ap.add_argument("--npz", type=int, default=127)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--hord", type=int, default=10, help="hord_mt = hord_vt = hord_tm = hord_dp (reference default 10)")
    ap.add_argument("--no-model-step", action="store_true", help="skip the SYPD (whole model step) leg")
    ap.add_argument("--nq", type=int, default=4, help="advected tracers in the SYPD leg")
    ap.add_argument("--model-step-multi", action="store_true", help="run the SYPD leg on N > 1 GPUs too")
    ap.add_argument("--general-metrics", action="store_true",
                    help="FV3_MI355X_GEOM=0: read every metric row (what a cubed-sphere gridstruct needs) instead of "
                         "using the uniform-Cartesian kernels the library selects for this doubly periodic gridstruct")
    return ap.parse_args()


def cpu_baseline(nx, seconds):
    """The oracle (oracle/libfvo.so: C restatement, OpenMP over k like dyn_core.F90:436,658) timed on
    this host's cores on a bounded sample of the same workload."""
    import oracle_lib as O
    import parity_common as P
    from fields import smooth_state
    from gfdl_atmos_cubed_sphere_amd.layout import Bounds, periodic_fill
    from test_oracle_properties import default_levels
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    npz = max(2, min(127, 2 * cores))
    bd = Bounds(1, nx, 1, nx)
    g = P.make_grid(bd, False)
    st = smooth_state(bd, npz, noise=0.05)
    f = {k: v for k, v in st.items()}
    for n, kind in P.CSW_OUT + (("mfx", "FX"), ("mfy", "FY"), ("cx", "CX"), ("cy", "CY"), ("crx", "CX"),
                                ("cry", "CY"), ("xfx", "CX"), ("yfx", "CY"), ("heat_source", "CC"),
                                ("diss_est", "CC")):
        f[n] = bd.zeros(kind, npz)
    par = dict(P.DSW_PAR)
    par.update(nord=1, nord_v=1, nord_w=1, nord_t=1, d2_bg=0., damp_v=0., damp_w=0., damp_t=0., d_con=0.,
               hydrostatic=0, use_cond=0)
    lev = default_levels(npz)
    keep = {k: f[k].copy(order="F") for k in ("delp", "pt", "u", "v", "w")}
    reps, t_used = 0, 0.0
    times = []
    while t_used < seconds or reps < 2:
        for k, v in keep.items():
            f[k][...] = v  # d_sw updates in place (reference semantics); restore the inputs
        t0 = time.perf_counter()
        O.c_sw_3d(g, npz, f, nord=1, dt2=3.0, hydrostatic=False)
        O.d_sw_3d(g, npz, par, lev, f)
        dt = time.perf_counter() - t0
        times.append(dt)
        t_used += dt
        reps += 1
        if reps >= 50:
            break
    best = float(np.median(times))
    return {"value": nx * nx * npz / best, "unit": "cell-updates/s", "cores": cores, "kind": "port",
            "sample": f"{nx}x{nx}x{npz} doubly periodic tile, c_sw+d_sw pair, median of {reps} reps "
                      f"({t_used:.1f} s CPU wall), OpenMP over k on {cores} threads"}


def model_step_leg(a, torch, dist, world, rank, px, py, bd, g, stream):
    """SYPD leg: whole nonhydrostatic model steps (fv_dynamics.F90:460-665 k_split loop: n_split acoustic substeps,
    tracer_2d, Lagrangian_to_Eulerian) on the same tile, dt_atmos=225 s, k_split=2, n_split=5 (C384 settings)."""
    import parity_dyn as D
    import parity_nh as N
    from gfdl_atmos_cubed_sphere_amd import lib as L
    from gfdl_atmos_cubed_sphere_amd.dyn_core import DynFlags
    from gfdl_atmos_cubed_sphere_amd.fv_dynamics import FvDynamics
    from gfdl_atmos_cubed_sphere_amd.layout import Bounds
    nx, npz, nq = a.nx, a.npz, a.nq
    ctx = L.Context(g, npz, stream=stream.cuda_stream)
    st, _ = D.make_state(Bounds(1, nx, 1, nx), npz)
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
    nrep = 3
    t0 = time.perf_counter()
    for _ in range(nrep):
        fv.step(dt_atmos)
    fence()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    wall = el / nrep
    w = fv.dc.d["w"].download()
    ctx.profile(True)
    fv.step(dt_atmos)
    rep = ctx.profile_report()
    ctx.profile(False)
    ctx.close()
    return {"sypd": dt_atmos / (365.0 * wall), "wall_s_per_dt_atmos": wall, "dt_atmos_s": dt_atmos, "k_split": k_split,
            "n_split": n_split, "nq": nq, "dx_m": 26000.0, "finite": bool(np.isfinite(w).all()),
            "note": f"one {nx}x{nx}x{npz} doubly periodic tile per GPU ({world} tile(s)); a C384 sphere is 6 such tiles, "
                    "so this is the SYPD of a 6-GPU one-face-per-GPU run before cube-edge exchange cost",
            "kernels_ms_per_dt_atmos": {k: round(v[1], 3) for k, v in rep.items()}}


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
    torch.cuda.set_device(local)
    # FV3_BENCH_LOOPBACK=1 (one GPU): every halo message goes through RCCL to this same rank and d_sw runs in its
    # interior / rest form -- the per-step flow of the N-GPU runs, to see what the message path costs
    loopback = world == 1 and os.environ.get("FV3_BENCH_LOOPBACK") == "1"
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    elif loopback:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))

    import parity_common as P
    from fields import smooth_state
    from gfdl_atmos_cubed_sphere_amd import lib as L
    from gfdl_atmos_cubed_sphere_amd.halo import HaloExchanger, choose_layout
    from gfdl_atmos_cubed_sphere_amd.layout import Bounds
    from test_oracle_properties import default_levels

    nx, npz = a.nx, a.npz
    px, py = choose_layout(world)
    # this rank's block of the (nx*px) x (nx*py) doubly periodic domain
    ix, iy = rank % px, rank // px
    bd = Bounds(1 + ix * nx, (ix + 1) * nx, 1 + iy * nx, (iy + 1) * nx)
    from gfdl_atmos_cubed_sphere_amd.grid import doubly_periodic
    g = doubly_periodic(bd, nx * px + 1, nx * py + 1, dx_const=26000.0, dy_const=26000.0)
    stream = torch.cuda.current_stream()
    if a.general_metrics:
        os.environ["FV3_MI355X_GEOM"] = "0"
    ctx = L.Context(g, npz, stream=stream.cuda_stream)
    geom = ctx.geom
    # FV3_BENCH_SPLIT=1: exercise the multi-rank flow (start / d_sw interior / finish / d_sw rest) on one GPU to see its cost
    halo = HaloExchanger(ctx, px, py, rank, world, split_single=loopback or os.environ.get("FV3_BENCH_SPLIT") == "1",
                         loopback=loopback)

    st = smooth_state(Bounds(1, nx, 1, nx), npz, noise=0.05)  # same synthetic block on every rank
    d = {k: ctx.from_host(v) for k, v in st.items()}
    del st
    for n, kind in P.CSW_OUT:
        d[n] = ctx.zeros(kind, npz)
    for n, kind in (("mfx", "FX"), ("mfy", "FY"), ("cx", "CX"), ("cy", "CY"), ("crx", "CX"), ("cry", "CY"),
                    ("xfx", "CX"), ("yfx", "CY"), ("delp_out", "A"), ("pt_out", "A"), ("u_out", "U"),
                    ("v_out", "V"), ("w_out", "A"), ("heat_s", "CC"), ("diss_e", "CC")):
        d[n] = ctx.zeros(kind, npz)
    ctx.dsw_levels(default_levels(npz))
    dt = 22.5   # C384 acoustic step: dt_atmos 225 s / k_split 2 / n_split 5
    par = dict(P.DSW_PAR)
    par.update(dt=dt, hydrostatic=0, use_cond=0, hord_mt=a.hord, hord_vt=a.hord, hord_tm=a.hord, hord_dp=a.hord)

    def step():
        ctx.c_sw(d["delpc"], d["delp"], d["ptc"], d["pt"], d["u"], d["v"], d["w"], d["uc"], d["vc"], d["ua"],
                 d["va"], d["wc"], d["ut"], d["vt"], d["divg_d"], 1, 0.5 * dt, False)
        dsw_args = (par, None, d["delp"], d["pt"], d["u"], d["v"], d["w"], d["uc"], d["vc"], d["ua"], d["va"],
                    d["divg_d"], d["mfx"], d["mfy"], d["cx"], d["cy"], d["crx"], d["cry"], d["xfx"], d["yfx"], None,
                    d["delp_out"], d["pt_out"], d["u_out"], d["v_out"], d["w_out"], None, d["heat_s"], d["diss_e"])
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

    def fence():
        torch.cuda.synchronize()
        if world > 1:
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
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    cells = nx * nx * npz
    value = cells * world * a.steps / el
    finite = bool(np.isfinite(d["u_out"].download()).all())

    # ---- per-kernel HIP-event timing (separate pass so the events do not perturb `value`) ----
    ctx.profile(True)
    nprof = max(3, min(10, a.steps))
    for _ in range(nprof):
        step()
    rep = ctx.profile_report()
    ctx.profile(False)
    per_kernel, launches = {}, {}
    grouped = {}
    for name, (n, ms) in rep.items():
        launches[name] = {"launches_per_step": n / nprof, "avg_ms": ms / n}
        gname = GROUP.get(name, name)
        grouped[gname] = grouped.get(gname, 0.0) + ms / nprof          # ms per step of the logical kernel
    for name, ms in grouped.items():
        if name in ALG_BYTES:
            avg = ms * 1e-3
            per_kernel[name] = {"avg_ms": ms, "GBps": cells * ALG_BYTES[name] / avg / 1e9,
                                "frac": cells * ALG_BYTES[name] / avg / HBM_PEAK}
    dom = max(per_kernel, key=lambda k: per_kernel[k]["avg_ms"]) if per_kernel else None
    t_pair = sum(v["avg_ms"] for v in per_kernel.values()) * 1e-3
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if dom and os.path.exists(tfile):
        traffic = json.load(open(tfile)).get(dom)
    roof = None
    if dom:
        ach = cells * ALG_BYTES[dom] / (per_kernel[dom]["avg_ms"] * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": ach / (HBM_PEAK / 1e9), "traffic": traffic,
                "alg_bytes_per_cell": ALG_BYTES[dom], "avg_ms": per_kernel[dom]["avg_ms"],
                "per_kernel": per_kernel, "launches": launches,
                "pair": {"alg_bytes_per_cell": PAIR_ALG_BYTES, "kernels_ms": t_pair * 1e3,
                         "frac": cells * PAIR_ALG_BYTES / t_pair / HBM_PEAK if t_pair > 0 else None,
                         # the same against the wall clock of the timed region (the sponge-level tile kernels overlap
                         # the marching kernels on a side stream there, so it can beat the sum of the launches)
                         "wall_ms": el / a.steps * 1e3,
                         "frac_wall": cells * PAIR_ALG_BYTES / (el / a.steps) / HBM_PEAK}}

    out = {"metric": "c_sw+d_sw cell-updates/s", "value": value, "unit": "cell-updates/s", "n_gpus": world,
           "steps": a.steps, "warmup": a.warmup, "ms_per_step": el / a.steps * 1e3, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"doubly periodic {nx}x{nx}x{npz} tile per GPU (C384L127-sized), nonhydrostatic, "
                                  f"c_sw+d_sw pair, hord {a.hord}/{a.hord}/{a.hord}/{a.hord}, nord=1, d4_bg=0.16",
                      "layout": f"{px}x{py}", "halo": ("RCCL send/recv (loopback)" if loopback else "periodic copy") if world == 1 else "RCCL send/recv",
                      # what fv3_grid_upload found in the metric arrays (fv3_grid_geom)
                      "gridstruct": {0: "general metric rows", 1: "orthogonal (angle terms not read)",
                                     2: "orthogonal + uniform (metric terms as scalars)"}[geom]},
           "finite": finite, "roofline": roof}
    ctx.close()
    ctx = None
    del d
    # the secondary legs must never cost the headline line: a failure there is reported, not raised
    out["model_step"] = None
    # N > 1: the SYPD leg is off unless asked for (a failure on one rank would leave the others in a collective)
    if not a.no_model_step and (world == 1 or a.model_step_multi):
        try:
            out["model_step"] = model_step_leg(a, torch, dist, world, rank, px, py, bd, g, stream)
        except Exception as e:  # noqa: BLE001
            out["model_step"] = {"error": f"{type(e).__name__}: {e}"}
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
