"""Worker of tests/test_gpu_parity.py::test_halo_messages_through_rccl_loopback: one process, one GPU, torch.distributed
backend "nccl" (= RCCL) with world_size 1.  The halo exchanger runs in loopback mode, i.e. every one of the 8 messages of
a field group is sent to and received from this rank through batch_isend_irecv -- the message path the N-GPU runs use
(pack kernel -> RCCL send/recv on its stream -> unpack kernel), checked against the periodic fill of the same fields."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import torch.distributed as dist

import parity_common as P
from gfdl_atmos_cubed_sphere_amd import lib as L
from gfdl_atmos_cubed_sphere_amd.halo import HaloExchanger
from gfdl_atmos_cubed_sphere_amd.layout import Bounds, periodic_fill


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29533")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    nx, ny, nk = 70, 41, 5
    bd = Bounds(1, nx, 1, ny)
    g = P.make_grid(bd, False)
    stream = torch.cuda.current_stream()
    ctx = L.Context(g, nk, stream=stream.cuda_stream)
    halo = HaloExchanger(ctx, 1, 1, 0, 1, loopback=True)
    rng = np.random.default_rng(3)
    fields, refs = [], []
    for kind in ("V", "U", "B", "A"):
        a = np.asfortranarray(rng.uniform(-1, 1, bd.shape(kind, nk)))
        ref = a.copy(order="F")
        for k in range(nk):
            periodic_fill(bd, ref[:, :, k], kind)
        fields.append((ctx.from_host(a), kind))
        refs.append(ref)
    for rep in range(3):                       # repeated use of the cached message buffers
        pend = halo.start(fields[:3], defer=(rep == 1))   # overlapped form: start [... post] ... finish
        halo.post(pend)
        halo.finish(pend)
        halo.update(fields[3:])
    ctx.sync()
    for (dev, kind), ref in zip(fields, refs):
        got = dev.download()
        assert np.array_equal(got, ref), f"halo of kind {kind} differs after the RCCL loopback exchange"
    ctx.close()
    # the substep loop with every group halo update as RCCL messages, the delp / pt and zh / pkc groups kept in flight across
    # update_dz_d + Riem_Solver3 and pe_halo / pk3_halo (DynCore.run with overlaps_groups): same state as the periodic copies
    import parity_dyn as D
    import parity_nh as N
    from gfdl_atmos_cubed_sphere_amd.dyn_core import DynCore, DynFlags
    nx, ny, npz = 40, 24, 8
    bd = Bounds(1, nx, 1, ny)
    g = P.make_grid(bd, False)
    st, dp0 = D.make_state(bd, npz)
    fl = DynFlags(n_split=3, ptop=N.PTOP)
    out = []
    for lb in (False, True):
        ctx = L.Context(g, npz, stream=stream.cuda_stream)
        h = HaloExchanger(ctx, 1, 1, 0, 1, loopback=True, split_single=True) if lb else None
        dc = DynCore(ctx, fl, dp0, halo=h)
        assert bool(getattr(dc.halo, "overlaps_groups", False)) == lb
        dc.set_state(st["u"], st["v"], st["w"], st["delp"], st["pt"], st["delz"], st["phis"])
        dc.run(6.0)
        out.append(dc.get_state())
        ctx.close()
    for n in ("u", "v", "w", "delp", "pt", "delz"):
        assert np.array_equal(out[0][n], out[1][n]), f"{n}: the lagged RCCL exchange changes the substeps"
    dist.destroy_process_group()
    print("rccl loopback ok")


if __name__ == "__main__":
    main()
