"""The exchange BEHIND THE C ABI (fv3_comm_init, fv3_halo_start / _complete, fv3_cube_halo_start / _complete, fv3_allreduce_max,
fv3_ordered_sum) with REAL rank counts: 2 / 4 processes on the doubly periodic domain, 6 processes x 1 face and 3 processes x 2 faces
on the cubed sphere.  The processes run the host-emulation build, whose exchange is the product's code -- pack lists, the order of the
sends and receives of a group, unpack -- over a file transport in the place of RCCL (csrc/fv3_api.hip, "the message transport of the
logic harness": the k-th message a -> b is matched with the k-th receive b posts for a and refused if its size differs, the matching
rule of ncclSend / ncclRecv inside a group).  What one GPU in loopback cannot falsify -- a sender and a receiver that disagree about
the order or the content of their messages -- fails here.  Reference: tools/fv_mp_mod.F90:498-546, :646-876,
model/dyn_core.F90:1151-1163."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _emu():
    from gfdl_atmos_cubed_sphere_amd.lib import Fv3Lib
    return Fv3Lib(os.path.join(HERE, "hostemu", "libfv3_hostemu.so"))


def _unique_id():
    sys.path.insert(0, os.path.dirname(HERE))
    emu = _emu()
    buf = (C.c_ubyte * 128)()
    emu.check(emu.dll.fv3_comm_get_unique_id(buf), "fv3_comm_get_unique_id")
    return bytes(buf)


def _setup():
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["OMP_NUM_THREADS"] = "2"


# ---- doubly periodic domain ------------------------------------------------------------------------------------------------------
def _periodic_worker(rank, world, uid, ok, case):
    _setup()
    import oracle_dyn_core as OD
    import parity_common as P
    import parity_dyn as D
    import parity_nh as N
    from gfdl_atmos_cubed_sphere_amd.dyn_core import DynCore, DynFlags
    from gfdl_atmos_cubed_sphere_amd.grid import doubly_periodic
    from gfdl_atmos_cubed_sphere_amd.halo import HaloExchanger, choose_layout
    from gfdl_atmos_cubed_sphere_amd.layout import Bounds
    from gfdl_atmos_cubed_sphere_amd.lib import Context
    from test_multirank_dyn_core import _block
    emu = _emu()
    px, py = choose_layout(world)
    ix, iy = rank % px, rank // px
    good = True
    if case == "halo":
        # every field kind of a group, the halos against the periodic continuation of a global analytic field
        nx, ny, npz = 9, 7, 3
        bd = Bounds(1 + ix * nx, (ix + 1) * nx, 1 + iy * ny, (iy + 1) * ny)
        ctx = Context(doubly_periodic(bd, nx * px + 1, ny * py + 1), npz, lib=emu)
        halo = HaloExchanger(ctx, px, py, rank, world, native=True, unique_id=uid)
        gx, gy = nx * px, ny * py

        def field(kind, seed):
            ilo, ihi, jlo, jhi = bd.limits(kind)
            i = np.arange(ilo, ihi + 1)[:, None, None]
            j = np.arange(jlo, jhi + 1)[None, :, None]
            k = np.arange(npz)[None, None, :]
            # the value of the OWNER of a point: staggered edge points ie+1 / je+1 belong to this block, everything else wraps
            return np.asfortranarray(seed + 1000.0 * ((i - 1) % gx) + ((j - 1) % gy) + 0.001 * k)
        for group in ([("A", 1.0)], [("V", 2.0), ("U", 3.0), ("B", 4.0)], [("A", 5.0), ("A", 6.0), ("U", 7.0), ("V", 8.0)]):
            full = [field(kind, seed) for kind, seed in group]
            dev = []
            for (kind, _), f in zip(group, full):
                g_ = np.full_like(f, np.nan)
                v = bd.view(g_, kind, *_compute(bd, kind))
                v[...] = bd.view(f, kind, *_compute(bd, kind))
                dev.append(ctx.from_host(g_))
            halo.update([(d, kind) for d, (kind, _) in zip(dev, group)])
            for d, f, (kind, _) in zip(dev, full, group):
                got = d.download()
                # the staggered edge of the high side is owned by the neighbour's compute domain in the analytic field: compare where
                # both agree by construction (every point the update fills, and the compute domain)
                if not np.array_equal(got, _expected(bd, kind, f, gx, gy)):
                    print("rank", rank, kind, "halo mismatch", flush=True)
                    good = False
        # mp_reduce_max and the reproducing sum through the context's communicator
        mx = ctx.allreduce_max(np.array([float(rank), -float(rank), 3.5]))
        good = good and list(mx) == [float(world - 1), 0.0, 3.5]
        rng = np.random.default_rng(9)
        a = rng.normal(0, 1e9, 3000) * rng.choice([1e-14, 1.0, 1e9], 3000)
        cuts = np.linspace(0, a.size, world + 1).astype(int)
        import math
        s = ctx.ordered_sum(a[cuts[rank]:cuts[rank + 1]])
        good = good and abs(s - math.fsum(a)) <= abs(s) * 2.3e-16
        ctx.close()
    else:
        # whole nonhydrostatic substeps: DynCore with every group halo update through fv3_halo_start / _complete
        nx, ny, npz = 12, 10, 6
        bd_g = Bounds(1, nx * px, 1, ny * py)
        st, dp0 = D.make_state(bd_g, npz)
        fl = DynFlags(n_split=2, ptop=N.PTOP)
        ref = OD.run(P.make_grid(bd_g, False), npz, fl, dp0, st, 4.0)
        bd = Bounds(1 + ix * nx, (ix + 1) * nx, 1 + iy * ny, (iy + 1) * ny)
        ctx = Context(doubly_periodic(bd, nx * px + 1, ny * py + 1), npz, lib=emu)
        dc = DynCore(ctx, fl, dp0, px=px, py=py, rank=rank, world=world,
                     halo=HaloExchanger(ctx, px, py, rank, world, native=True, unique_id=uid))
        loc = {n: _block(st[n], k, bd_g, bd) for n, k in (("u", "U"), ("v", "V"), ("w", "A"), ("delp", "A"), ("pt", "A"), ("phis", "A"))}
        delz = np.asfortranarray(st["delz"][ix * nx:(ix + 1) * nx, iy * ny:(iy + 1) * ny, :].copy())
        dc.set_state(loc["u"], loc["v"], loc["w"], loc["delp"], loc["pt"], delz, loc["phis"])
        dc.run(4.0)
        got = dc.get_state()
        for n, kind, rr in (("u", "U", (bd.is_, bd.ie, bd.js, bd.je + 1)), ("v", "V", (bd.is_, bd.ie + 1, bd.js, bd.je)),
                            ("w", "A", (bd.is_, bd.ie, bd.js, bd.je)), ("delp", "A", (bd.is_, bd.ie, bd.js, bd.je)),
                            ("pt", "A", (bd.is_, bd.ie, bd.js, bd.je)), ("zh", "A", (bd.is_, bd.ie, bd.js, bd.je))):
            e = P.rel_rms(bd.view(got[n], kind, *rr), bd_g.view(ref[n], kind, *rr))
            if not (e <= 1e-13):
                print("rank", rank, n, e, flush=True)
                good = False
        ctx.close()
    ok[rank] = 1 if good else 0


def _compute(bd, kind):
    si, sj = {"A": (0, 0), "U": (0, 1), "V": (1, 0), "B": (1, 1)}[kind]
    return bd.is_, bd.ie + si, bd.js, bd.je + sj


def _expected(bd, kind, full, gx, gy):
    """the analytic field is periodic in the CELL index; on a staggered kind the point ie+1 of this block is its own compute point and
    equals the analytic value there, so the filled array is the analytic field everywhere"""
    return full


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("case", ["halo", "substeps"])
def test_periodic_exchange_behind_the_c_abi(world, case):
    subprocess.check_call(["make", "-C", os.path.join(HERE, "hostemu"), "-s"])
    uid = _unique_id()
    ok = mp.get_context("spawn").Array("i", [0] * world)
    mp.spawn(_periodic_worker, args=(world, uid, ok, case), nprocs=world, join=True)
    assert list(ok) == [1] * world


# ---- cubed sphere: 6 x 1 face, 3 x 2 faces, 2 x 3 faces ---------------------------------------------------------------------------
def _cube_worker(rank, world, uid, ok, case):
    _setup()
    import cubed_common as CC
    import parity_common as P
    import parity_cubed as PC
    from gfdl_atmos_cubed_sphere_amd.cubed_dyn import CubeNativeAdapter, MultiContext
    from gfdl_atmos_cubed_sphere_amd.dyn_core import DynFlags
    from gfdl_atmos_cubed_sphere_amd.fv_dynamics import FvDynamics
    from gfdl_atmos_cubed_sphere_amd.lib import Context
    emu = _emu()
    npx, npz = 13, 4
    per = 6 // world
    faces = list(range(rank * per, (rank + 1) * per))
    face_rank = [t // per for t in range(6)]
    cs, gs = CC.sphere(npx)
    bd = gs[0].bd
    good = True
    ctxs = [Context(gs[t], npz, lib=emu) for t in faces]
    mctx = MultiContext(ctxs) if per > 1 else ctxs[0]
    halo = CubeNativeAdapter(mctx, faces, face_rank, rank, world, uid)
    if case == "halo":
        rng = np.random.default_rng(7)          # the same global fields on every rank
        for kind, kinds in (("A", ("A",)), ("A2", ("A", "A")), ("B", ("B",)), ("D", ("U", "V")), ("C", ("V", "U")), ("Dedge", ("U", "V"))):
            host = [[np.asfortranarray(rng.uniform(-1, 1, bd.shape(k, npz))) for _ in range(6)] for k in kinds]
            ref = [[x.copy(order="F") for x in a] for a in host]
            if kind == "A2":
                cs.topo.update("A", ref[0]); cs.topo.update("A", ref[1])
            else:
                cs.topo.update(kind, ref[0] if len(kinds) == 1 else (ref[0], ref[1]))
            if per > 1:
                dev = [mctx.from_host([a[t] for t in faces]) for a in host]
            else:
                dev = [mctx.from_host(a[faces[0]]) for a in host]
            if kind == "Dedge":
                halo.sync_edges(dev[0], dev[1])
            else:
                halo.update(list(zip(dev, kinds)))
            for m in range(len(kinds)):
                got = dev[m].download()
                got = got if per > 1 else [got]
                for n, t in enumerate(faces):
                    if not np.array_equal(got[n], ref[m][t]):
                        print("rank", rank, kind, m, "face", t, "halo mismatch", flush=True)
                        good = False
    else:
        # a whole nonhydrostatic fv_dynamics step (substeps, tracer_2d with its reduced Courant maximum, remap) against the six-face oracle
        class _Reduce:      # torch.distributed's surface as tracer2d / global_sum use it, on the context's communicator
            class ReduceOp:
                MAX, SUM = "max", "sum"

            def __init__(self, ctx, world):
                self.ctx, self.world = ctx, world

            def get_backend(self):
                return "fv3"

            def is_initialized(self):
                return True

            def get_world_size(self):
                return self.world

            def all_reduce(self, t, op=None):
                a = t.numpy()
                assert op == "max", "only mp_reduce_max is routed here"
                a[...] = self.ctx.allreduce_max(a.astype(np.float64)).reshape(a.shape)
        cs, gs, st = CC.nh_state(npx, npz + 1)
        km = npz + 1
        for c_ in ctxs:
            c_.close()
        ctxs = [Context(gs[t], km, lib=emu) for t in faces]
        mctx = MultiContext(ctxs) if per > 1 else ctxs[0]
        halo = CubeNativeAdapter(mctx, faces, face_rank, rank, world, uid)
        fl = DynFlags(n_split=2, hydrostatic=False)
        sig = np.linspace(0.0, 1.0, km + 1) ** 1.5
        ak, bk = fl.ptop * (1.0 - sig), sig.copy()
        dp0 = (ak[1:] - ak[:-1]) + (bk[1:] - bk[:-1]) * 1.0e5
        nq = 2
        q0 = PC.tracer_fields(cs, km, nq)
        fv = FvDynamics(mctx, fl, ak, bk, nq=nq, k_split=2, halo=halo, dist=_Reduce(ctxs[0], world))
        ref = CC.oracle_fv_step_nh(cs, gs, fl, dp0, st, ak, bk, 600.0, 2, fv.remap_par, km, q=q0)
        pick = (lambda name: [st[t][name] for t in faces]) if per > 1 else (lambda name: st[faces[0]][name])
        fv.dc.set_state(pick("u"), pick("v"), pick("w"), pick("delp"), pick("pt"), pick("delz"), pick("phis"))
        fv.set_tracers([q0[t] for t in faces] if per > 1 else q0[faces[0]])
        fv.step(600.0)
        d = fv.dc.d
        r = (bd.is_, bd.ie, bd.js, bd.je)
        for n, kind, rr in (("u", "U", (bd.is_, bd.ie, bd.js, bd.je + 1)), ("v", "V", (bd.is_, bd.ie + 1, bd.js, bd.je)), ("delp", "A", r),
                            ("pt", "A", r), ("w", "A", r)):
            got = d[n].download()
            got = got if per > 1 else [got]
            for m, t in enumerate(faces):
                e = P.rel_rms(bd.view(got[m], kind, *rr), bd.view(ref[t][n], kind, *rr))
                if not (e <= 1e-13):
                    print("rank", rank, "face", t, n, e, flush=True)
                    good = False
    for c_ in ctxs:
        c_.close()
    ok[rank] = 1 if good else 0


@pytest.mark.parametrize("world", [6, 3, 2])
@pytest.mark.parametrize("case", ["halo", "step"])
def test_cube_edge_exchange_behind_the_c_abi(world, case):
    """one face per process (BASELINE config 5's layout), two and three faces per process (a sphere on 3 / 2 GPUs): every message
    through fv3_cube_halo_start / _complete"""
    if case == "step" and world == 2:
        pytest.skip("covered by 6 x 1 and 3 x 2")
    subprocess.check_call(["make", "-C", os.path.join(HERE, "hostemu"), "-s"])
    uid = _unique_id()
    ok = mp.get_context("spawn").Array("i", [0] * world)
    mp.spawn(_cube_worker, args=(world, uid, ok, case), nprocs=world, join=True)
    assert list(ok) == [1] * world
