"""One cubed-sphere face per rank (BASELINE configs[4] in small): six gloo ranks on CPU, each with ONE context of the
host-emulation build, the product's DynCore / FvDynamics orchestration and the CubeHaloRank message exchange (pack gather ->
grouped send/recv -> unpack gather); every rank's face must equal the single-process six-face oracle.  On six GPUs the same
code runs over RCCL."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, ok, case):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cubed_common as CC
        import parity_common as P
        import parity_cubed as PC
        from gfdl_atmos_cubed_sphere_amd.cubed_dyn import CubeRankAdapter
        from gfdl_atmos_cubed_sphere_amd.dyn_core import DynCore, DynFlags
        from gfdl_atmos_cubed_sphere_amd.fv_dynamics import FvDynamics
        from gfdl_atmos_cubed_sphere_amd.lib import Context, Fv3Lib
        emu = Fv3Lib(os.path.join(HERE, "hostemu", "libfv3_hostemu.so"))
        npx, npz = 13, 5
        good = True
        t = rank
        if case == "halo":
            # every field kind through the message path against the numpy application of the same tables
            cs, gs = CC.sphere(npx)
            ctx = Context(gs[t], npz, lib=emu)
            halo = CubeRankAdapter(ctx, t, npx, dist, topo=CC.product_topo(npx))
            bd = gs[0].bd
            rng = np.random.default_rng(7)          # the same global fields on every rank
            for kind, kinds in (("A", ("A",)), ("A2", ("A", "A")), ("B", ("B",)), ("D", ("U", "V")), ("C", ("V", "U")), ("Dedge", ("U", "V"))):
                host = [[np.asfortranarray(rng.uniform(-1, 1, bd.shape(k, npz))) for _ in range(6)] for k in kinds]
                ref = [[x.copy(order="F") for x in a] for a in host]
                if kind == "A2":
                    cs.topo.update("A", ref[0]); cs.topo.update("A", ref[1])
                else:
                    cs.topo.update(kind, ref[0] if len(kinds) == 1 else (ref[0], ref[1]))
                dev = [ctx.from_host(a[t]) for a in host]
                if kind == "Dedge":
                    halo.sync_edges(dev[0], dev[1])
                else:
                    halo.update(list(zip(dev, kinds)))
                for m in range(len(kinds)):
                    if not np.array_equal(dev[m].download(), ref[m][t]):
                        print("rank", rank, kind, m, "halo mismatch", flush=True)
                        good = False
            ctx.close()
        elif case == "consv":
            # the energy fixer with one face per rank: te_2d / zsum0 of the six faces summed through the all-reduced integer digits
            r_ = PC.check_jw_consv(emu, npx=npx, npz=8, face=t, dist=dist)
            good = max(r_.values()) <= 1e-12
        else:
            hydro = case == "hydro"
            cs, gs, st = CC.hydro_state(npx, npz) if hydro else CC.nh_state(npx, npz)
            fl = DynFlags(n_split=2, hydrostatic=hydro, **(dict(d_ext=0.0) if hydro else {}))
            sig = np.linspace(0.0, 1.0, npz + 1) ** 1.5
            ak, bk = fl.ptop * (1.0 - sig), sig.copy()
            dp0 = (ak[1:] - ak[:-1]) + (bk[1:] - bk[:-1]) * 1.0e5          # dp_ref as fv_dynamics builds it (dyn_core.F90:241-244)
            nq = 2
            q0 = PC.tracer_fields(cs, npz, nq)
            ctx = Context(gs[t], npz, lib=emu)
            fv = FvDynamics(ctx, fl, ak, bk, nq=nq, k_split=2, halo=CubeRankAdapter(ctx, t, npx, dist, topo=CC.product_topo(npx)), dist=dist)
            if hydro:
                ref = CC.oracle_fv_step_hydro(cs, gs, fl, st, ak, bk, 600.0, 2, fv.remap_par, npz, q=q0)
            else:
                ref = CC.oracle_fv_step_nh(cs, gs, fl, dp0, st, ak, bk, 600.0, 2, fv.remap_par, npz, q=q0)
            s = st[t]
            bd = gs[0].bd
            z = np.zeros_like(s["delp"]) if hydro else s["w"]
            dz = bd.zeros("CC", npz) if hydro else s["delz"]
            fv.dc.set_state(s["u"], s["v"], z, s["delp"], s["pt"], dz, s["phis"])
            fv.set_tracers(q0[t])
            fv.step(600.0)
            d = fv.dc.d
            r = (bd.is_, bd.ie, bd.js, bd.je)
            names = [("u", "U", (bd.is_, bd.ie, bd.js, bd.je + 1)), ("v", "V", (bd.is_, bd.ie + 1, bd.js, bd.je)), ("delp", "A", r), ("pt", "A", r)]
            if not hydro:
                names.append(("w", "A", r))
            for n, kind, rr in names:
                e = P.rel_rms(bd.view(d[n].download(), kind, *rr), bd.view(ref[t][n], kind, *rr))
                if not (e <= 1e-13):
                    print("rank", rank, n, e, flush=True)
                    good = False
            got = d["q"].download()
            for iq in range(nq):
                e = P.rel_rms(bd.view(got[:, :, :, iq], "A", *r), bd.view(ref[t]["q"][:, :, :, iq], "A", *r))
                if not (e <= 1e-13):
                    print("rank", rank, "q", iq, e, flush=True)
                    good = False
            ctx.close()
        ok[rank] = 1 if good else 0
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["halo", "hydro", "nh", "consv"])
def test_one_face_per_rank_matches_the_six_face_oracle(case):
    subprocess.check_call(["make", "-C", os.path.join(HERE, "hostemu"), "-s"])
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world = 6
    ok = mp.get_context("spawn").Array("i", [0] * world)
    mp.spawn(_worker, args=(world, port, ok, case), nprocs=world, join=True)
    assert list(ok) == [1] * world
