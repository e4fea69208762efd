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
    dist.destroy_process_group()
    print("rccl loopback ok")


if __name__ == "__main__":
    main()
