"""Host side of ``tracer_2d`` (model/fv_tracer2d.F90:297-557): the part of the routine that needs a cross-rank
reduction and integer bookkeeping stays on the host, the arithmetic runs in the HIP kernels
(csrc/tracer_kernels.h)."""
from __future__ import annotations

import numpy as np


def tracer_2d(ctx, halo, q, q_nxt, dp1, dp1_nxt, mfx, mfy, cx, cy, xfx, yfx, nq: int, hord: int, q_split: int = 0,
              nord_tr: int = 0, trdm: float = 0.0, dist=None):
    """q/q_nxt: DeviceArray A x npz x nq ping-pong pair; dp1/dp1_nxt: A x npz pair; xfx, yfx: CX/CY work arrays.
    Returns (q, dp1, nsplt): the buffers that hold the result."""
    npz = ctx.npz
    cmax = ctx.tracer_2d_prep(q_split, cx, cy, xfx, yfx)                      # :362-400
    if isinstance(cmax, list):       # several domains in this process (the six faces): mp_reduce_max over them, :405
        cmax = np.max(np.stack(cmax), axis=0)
    if q_split == 0:
        if dist is None and getattr(halo, "world", 1) > 1:
            # without the reduction the ranks would sub-cycle differently and post mismatching q exchanges
            raise ValueError("tracer_2d on several ranks needs the process group (dist=) for mp_reduce_max(cmax)")
        if dist is not None:                                                   # mp_reduce_max(cmax, npz), :405
            import torch
            t = torch.from_numpy(cmax.copy())
            if dist.get_backend() == "nccl":
                t = t.cuda()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            cmax = t.cpu().numpy()
        c_global = float(np.max(cmax)) if npz != 1 else float(cmax[0])          # :407-412
        nsplt = int(1.0 + c_global)
    else:
        nsplt = q_split
    if nsplt != 1:                                                             # :421-456
        ksplt = (1.0 + cmax).astype(np.int32)
        frac = 1.0 / ksplt.astype(np.float64)
        ctx.tracer_2d_scale(frac, cx, xfx, mfx, cy, yfx, mfy)
    else:
        ksplt = np.ones(npz, dtype=np.int32)
    if trdm > 1.0e-4:
        halo.update([(dp1, "A")])                                              # dp1_pack, :466
    for it in range(1, nsplt + 1):                                             # :471-541
        halo.update([(q, "A")])                                                # q_pack, :474 / :536
        ctx.tracer_2d_step(it, nsplt, ksplt, nq, hord, nord_tr, trdm, q, q_nxt, dp1, dp1_nxt, mfx, mfy, cx, cy, xfx, yfx)
        q, q_nxt = q_nxt, q
        if it != nsplt:
            dp1, dp1_nxt = dp1_nxt, dp1
    return q, dp1, nsplt
