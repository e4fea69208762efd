"""The N>1 path of the substep loop: two ranks (gloo, CPU) each own half of a doubly periodic domain and run the
product's DynCore orchestration + halo exchange on the host-emulation build of the kernels; the union of their
blocks must equal the oracle's single-domain result.  (On GPUs the same code runs over RCCL.)"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _block(a, kind, bd_g, bd_l):
    """slice the local block (with halo) out of a global halo'd array"""
    from gfdl_atmos_cubed_sphere_amd.layout import Bounds
    ilo_g, _, jlo_g, _ = bd_g.limits(kind)
    ilo, ihi, jlo, jhi = bd_l.limits(kind)
    return np.asfortranarray(a[ilo - ilo_g:ihi - ilo_g + 1, jlo - jlo_g:jhi - jlo_g + 1].copy())


def _worker(rank, world, port, ok, nx=12, ny=10, npz=6, tj_fused=None, hydrostatic=False):
    if tj_fused:
        os.environ["FV3_MI355X_MARCH_TJ_FUSED"] = str(tj_fused)   # short segments: several per block
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle_dyn_core as OD
        import parity_common as P
        import parity_dyn as D
        import parity_nh as N
        from gfdl_atmos_cubed_sphere_amd.dyn_core import DynCore, DynFlags
        from gfdl_atmos_cubed_sphere_amd.grid import doubly_periodic
        from gfdl_atmos_cubed_sphere_amd.halo import choose_layout
        from gfdl_atmos_cubed_sphere_amd.layout import Bounds
        from gfdl_atmos_cubed_sphere_amd.lib import Context, Fv3Lib
        emu = Fv3Lib(os.path.join(HERE, "hostemu", "libfv3_hostemu.so"))
        px, py = choose_layout(world)
        bd_g = Bounds(1, nx * px, 1, ny * py)
        g_g = P.make_grid(bd_g, False)
        st, dp0 = D.make_state(bd_g, npz)
        fl = DynFlags(n_split=2, ptop=N.PTOP, hydrostatic=hydrostatic)
        ref = OD.run_hydrostatic(g_g, npz, fl, st, 4.0) if hydrostatic else OD.run(g_g, npz, fl, dp0, st, 4.0)
        ix, iy = rank % px, rank // px
        bd = Bounds(1 + ix * nx, (ix + 1) * nx, 1 + iy * ny, (iy + 1) * ny)
        g = doubly_periodic(bd, nx * px + 1, ny * py + 1)
        ctx = Context(g, npz, lib=emu)
        dc = DynCore(ctx, fl, dp0, px=px, py=py, rank=rank, world=world)
        loc = {n: _block(st[n], k, bd_g, bd) for n, k in (("u", "U"), ("v", "V"), ("w", "A"), ("delp", "A"), ("pt", "A"),
                                                         ("phis", "A"))}
        delz = np.asfortranarray(st["delz"][ix * nx:(ix + 1) * nx, iy * ny:(iy + 1) * ny, :].copy())
        dc.set_state(loc["u"], loc["v"], loc["w"], loc["delp"], loc["pt"], delz, loc["phis"])
        dc.run(4.0)
        got = dc.get_state()
        good = True
        names = (("u", "U", (bd.is_, bd.ie, bd.js, bd.je + 1)), ("v", "V", (bd.is_, bd.ie + 1, bd.js, bd.je)),
                 ("w", "A", (bd.is_, bd.ie, bd.js, bd.je)), ("delp", "A", (bd.is_, bd.ie, bd.js, bd.je)),
                 ("pt", "A", (bd.is_, bd.ie, bd.js, bd.je)), ("zh", "A", (bd.is_, bd.ie, bd.js, bd.je)))
        for n, kind, rr in (names[:2] + names[3:5] if hydrostatic else names):
            a = bd.view(got[n], kind, *rr)
            b = bd_g.view(ref[n], kind, *rr)
            e = P.rel_rms(a, b)
            if not (e <= 1e-13):
                print("rank", rank, n, e, flush=True)
                good = False
        ok[rank] = 1 if good else 0
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_dyn_core_two_ranks_match_single_domain(world):
    subprocess.check_call(["make", "-C", os.path.join(HERE, "hostemu"), "-s"])
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ok = mp.get_context("spawn").Array("i", [0] * world)
    mp.spawn(_worker, args=(world, port, ok), nprocs=world, join=True)
    assert list(ok) == [1] * world


def test_two_ranks_with_exchange_overlap_split():
    """blocks of 3 x 3 strips/segments per rank: d_sw's interior box runs between start and finish of the uc/vc
    exchange, the frame afterwards (DynCore.run); the result must still equal the single-domain oracle"""
    subprocess.check_call(["make", "-C", os.path.join(HERE, "hostemu"), "-s"])
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world = 2
    ok = mp.get_context("spawn").Array("i", [0] * world)
    mp.spawn(_worker, args=(world, port, ok, 120, 22, 3, 8), nprocs=world, join=True)
    assert list(ok) == [1] * world


def _sum_worker(rank, world, port, ok):
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import math
        from gfdl_atmos_cubed_sphere_amd.global_sum import reproducing_sum
        rng = np.random.default_rng(9)
        a = rng.normal(0, 1e9, 30000) * rng.choice([1e-14, 1.0, 1e9], 30000)        # the same array on every rank
        cuts = [0, 11111, 30000] if world == 2 else [0, 5000, 12000, 29000, 30000]
        s = reproducing_sum([a[cuts[rank]:cuts[rank + 1]]], dist)
        assert s == reproducing_sum([a]) and abs(s - math.fsum(a)) <= abs(s) * 2.3e-16, (s, math.fsum(a))
        ok[rank] = 1
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_reproducing_sum_does_not_depend_on_the_rank_layout(world):
    """g_sum(reproduce=.true.) of the energy fixer: the digits of the extended-fixed-point sum all-reduced as integers -- the sum
    over 2 or 4 ranks of differently sized pieces equals the one-rank sum bit for bit"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ok = mp.get_context("spawn").Array("i", [0] * world)
    mp.spawn(_sum_worker, args=(world, port, ok), nprocs=world, join=True)
    assert list(ok) == [1] * world


def test_hydrostatic_substeps_on_two_ranks_with_the_overlap_split():
    """the hydrostatic branch of the loop on two ranks (geopk, external-mode damping, one_grad_p), d_sw split around the uc / vc
    exchange: the union of the blocks equals the single-domain oracle"""
    subprocess.check_call(["make", "-C", os.path.join(HERE, "hostemu"), "-s"])
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world = 2
    ok = mp.get_context("spawn").Array("i", [0] * world)
    mp.spawn(_worker, args=(world, port, ok, 120, 22, 3, 8, True), nprocs=world, join=True)
    assert list(ok) == [1] * world
