"""Device-side halo updates of the six cubed-sphere faces held on ONE GPU (one context per face): the tables of
``cubed_sphere.CubeTopology`` uploaded once per field kind, one gather launch per update (``fv3_gather_run``).

Reference: ``mpp_update_domains`` / ``start_group_halo_update`` on the cubed-sphere mosaic (tools/fv_mp_mod.F90:498-546,
:646-876) with position CENTER / CORNER and gridtype DGRID_NE / CGRID_NE, ``mpp_get_boundary`` (model/dyn_core.F90:1151-
1163).  With one face per GPU the same tables split by source face into pack lists (message buffers) and unpack lists.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .cubed_sphere import CubeTopology

_vp = C.c_void_p


class CubeHalo:
    def __init__(self, ctxs, npx: int, ng: int = 3, topo: CubeTopology | None = None):
        assert len(ctxs) == 6
        self.ctxs = list(ctxs)
        self.ctx = ctxs[0]                  # all six faces live on this context's device; it launches the gathers
        self.lib = self.ctx.lib
        self.topo = topo or CubeTopology(npx, ng)
        self._handles = {}

    def _handle(self, kind: str, vector: bool):
        key = (kind, vector)
        if key in self._handles:
            return self._handles[key]
        tab = self.topo.boundary_table() if kind == "Dedge" else self.topo.table(kind)
        cols = [[], [], [], [], []]
        for t in range(6):
            for m, tb in enumerate(tab[t]):
                n = tb["dst"].size
                src_m = np.where(tb["comp"] == 0, m, 1 - m)
                cols[0].append(np.full(n, m * 6 + t))
                cols[1].append(tb["dst"])
                cols[2].append(src_m * 6 + tb["tile"])
                cols[3].append(tb["src"])
                cols[4].append(tb["sign"] if vector else np.ones(n, dtype=np.int64))
        arrs = [np.ascontiguousarray(np.concatenate(c), dtype=np.int32) for c in cols]
        h = _vp()
        ip = C.POINTER(C.c_int)
        self.lib.check(self.lib.dll.fv3_gather_create(self.ctx.h, C.c_int(arrs[0].size), *[a.ctypes.data_as(ip) for a in arrs],
                                                     C.byref(h)), "fv3_gather_create")
        self._handles[key] = h
        return h

    def update(self, kind: str, fields, vector: bool = True):
        """kind 'A' / 'B': fields = list of 6 DeviceArrays; 'D' / 'C' / 'Dedge': (list of 6, list of 6) = the two members of
        the pair (u, v resp. uc, vc).  vector=False: SCALAR_PAIR."""
        pair = kind in ("D", "C", "Dedge")
        mem = fields if pair else (fields,)
        ptrs = [d.ptr for lst in mem for d in lst]
        shapes = [d.shape for lst in mem for d in lst]
        nk = 1 if len(shapes[0]) == 2 else int(np.prod(shapes[0][2:]))     # (i, j, k) or the tracer array (i, j, k, iq)
        strides = [int(s[0] * s[1]) for s in shapes]
        n = len(ptrs)
        parr = (C.c_void_p * n)(*ptrs)
        sarr = (C.c_size_t * n)(*strides)
        self.lib.check(self.lib.dll.fv3_gather_run(self.ctx.h, self._handle(kind, vector), C.c_int(nk), C.c_int(n), parr, sarr),
                       "fv3_gather_run")

    def close(self):
        for h in self._handles.values():
            self.lib.dll.fv3_gather_destroy(h)
        self._handles = {}
