"""Halo exchange: single-rank periodic semantics and the multi-rank path over gloo (world_size 2 and 4
on CPU) against a global periodic array."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gfdl_atmos_cubed_sphere_amd.halo import HaloTopology, choose_layout, exchange_tensors
from gfdl_atmos_cubed_sphere_amd.layout import Bounds, periodic_fill

_STAG = {"A": (0, 0), "U": (0, 1), "V": (1, 0), "B": (1, 1)}


def test_choose_layout():
    assert choose_layout(1) == (1, 1) and choose_layout(2) == (2, 1)
    assert choose_layout(4) == (2, 2) and choose_layout(8) == (4, 2) and choose_layout(6) == (3, 2)


def _global_field(nxg, nyg, nk, kind, seed):
    """values defined by global (periodic) index so every rank can build its expected halo"""
    rng = np.random.default_rng(seed)
    return rng.uniform(-1, 1, (nxg, nyg, nk))


def _local_expected(G, bd, kind, nxg, nyg):
    ilo, ihi, jlo, jhi = bd.limits(kind)
    ii = (np.arange(ilo, ihi + 1) - 1) % nxg
    jj = (np.arange(jlo, jhi + 1) - 1) % nyg
    return G[np.ix_(ii, jj)]


@pytest.mark.parametrize("kind", ["A", "U", "V", "B"])
def test_single_rank_matches_periodic_fill(kind):
    bd = Bounds(1, 9, 1, 7)
    G = _global_field(9, 7, 2, kind, 3)
    exp = _local_expected(G, bd, kind, 9, 7)
    a = np.asfortranarray(exp.copy())
    # wipe the halo (keep compute points incl. the staggered edge)
    si, sj = _STAG[kind]
    mask = np.zeros(a.shape[:2], bool)
    mask[3:3 + 9 + si, 3:3 + 7 + sj] = True
    a[~mask] = np.nan
    t = torch.from_numpy(a)
    exchange_tensors(HaloTopology(bd, 1, 1, 0), [(t, kind)])
    np.testing.assert_array_equal(a, exp)
    b = np.asfortranarray(exp.copy())
    b[~mask] = np.nan
    for k in range(2):
        periodic_fill(bd, b[:, :, k], kind)
    np.testing.assert_array_equal(b, exp)


def _worker(rank, world, port, nx, ny, nk, ok):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        px, py = choose_layout(world)
        ix, iy = rank % px, rank // px
        bd = Bounds(1 + ix * nx, (ix + 1) * nx, 1 + iy * ny, (iy + 1) * ny)
        topo = HaloTopology(bd, px, py, rank)
        fields, exps = [], []
        for n, kind in enumerate(["A", "U", "V", "B", "A"]):
            G = _global_field(nx * px, ny * py, nk, kind, 10 + n)
            exp = _local_expected(G, bd, kind, nx * px, ny * py)
            a = np.asfortranarray(exp.copy())
            si, sj = _STAG[kind]
            mask = np.zeros(a.shape[:2], bool)
            mask[3:3 + nx + si, 3:3 + ny + sj] = True
            a[~mask] = -777.0
            fields.append((torch.from_numpy(a), kind))
            exps.append((a, exp))
        exchange_tensors(topo, fields[:2], dist)   # two packs, like the reference's grouped updates
        exchange_tensors(topo, fields[2:], dist)
        good = all(np.array_equal(a, e) for a, e in exps)
        ok[rank] = 1 if good else 0
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_multi_rank_gloo(world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ok = mp.get_context("spawn").Array("i", [0] * world)
    mp.spawn(_worker, args=(world, port, 8, 6, 3, ok), nprocs=world, join=True)
    assert list(ok) == [1] * world


def _worker_packed(rank, world, port, nx, ny, nk, ok, so):
    """the product path: pack kernel -> 8 grouped messages -> unpack kernel (host-emulation build of the kernels)"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gfdl_atmos_cubed_sphere_amd.grid import doubly_periodic
        from gfdl_atmos_cubed_sphere_amd.halo import HaloExchanger
        from gfdl_atmos_cubed_sphere_amd.lib import Context, Fv3Lib
        px, py = choose_layout(world)
        ix, iy = rank % px, rank // px
        bd = Bounds(1 + ix * nx, (ix + 1) * nx, 1 + iy * ny, (iy + 1) * ny)
        ctx = Context(doubly_periodic(bd, nx * px + 1, ny * py + 1), nk, lib=Fv3Lib(so))
        halo = HaloExchanger(ctx, px, py, rank, world)
        fields, exps = [], []
        for n, kind in enumerate(["A", "U", "V", "B", "A"]):
            levels = nk + (1 if n == 4 else 0)          # mixed level counts in one group
            G = _global_field(nx * px, ny * py, levels, kind, 10 + n)
            exp = _local_expected(G, bd, kind, nx * px, ny * py)
            a = np.asfortranarray(exp.copy())
            si, sj = _STAG[kind]
            mask = np.zeros(a.shape[:2], bool)
            mask[3:3 + nx + si, 3:3 + ny + sj] = True
            a[~mask] = -777.0
            fields.append((ctx.from_host(a), kind))
            exps.append(exp)
        halo.update(fields[:2])
        halo.update(fields[2:])
        halo.update(fields[:2])                         # cached group buffers are reused
        good = all(np.array_equal(f.download(), e) for (f, _), e in zip(fields, exps))
        ok[rank] = 1 if good else 0
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_multi_rank_packed_exchange(world):
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    subprocess.check_call(["make", "-s", "-C", os.path.join(here, "hostemu")])
    so = os.path.join(here, "hostemu", "libfv3_hostemu.so")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ok = mp.get_context("spawn").Array("i", [0] * world)
    mp.spawn(_worker_packed, args=(world, port, 8, 6, 3, ok, so), nprocs=world, join=True)
    assert list(ok) == [1] * world
