"""The six faces of the cubed sphere on ONE GPU behind the single-domain host code: ``MultiContext`` looks like one
``lib.Context`` (every kernel call is issued once per face on that face's context, every field is a ``FaceSet`` of six device
arrays) and ``CubeHaloAdapter`` looks like one ``halo.HaloExchanger`` (group halo updates -> the table-driven gathers of
``cubed_halo.CubeHalo``), so that ``dyn_core.DynCore`` / ``fv_dynamics.FvDynamics`` drive a whole sphere unchanged
(BASELINE configs 2: C96L79 on one MI355X).  With one face per GPU each rank holds one context and the adapter's tables
become pack / unpack lists around RCCL messages.
"""
from __future__ import annotations

import numpy as np

from .cubed_halo import CubeHalo, CubeHaloNative, CubeHaloRank
from .cubed_sphere import CubedSphere
from .lib import Context, FaceGroup


class FaceSet:
    """six DeviceArrays (one per face) handled as one field"""

    def __init__(self, arrs):
        self.a = list(arrs)
        self.shape = self.a[0].shape

    def __getitem__(self, t):
        return self.a[t]

    def upload(self, hosts):
        for a, h in zip(self.a, hosts if isinstance(hosts, (list, tuple)) else [hosts] * 6):
            a.upload(h)
        return self

    def download(self):
        return [a.download() for a in self.a]

    def zero(self):
        for a in self.a:
            a.zero()
        return self

    def copy_from(self, other):
        for a, o in zip(self.a, other.a):
            a.copy_from(o)
        return self


class MultiContext:
    """`group` (default: on, FV3_MI355X_FACE_GROUP=0 turns it off): the contexts form an fv3_group -- every kernel the faces issue in
    turn runs as ONE launch over all of them (lib.FaceGroup); the faces then share the first context's stream."""

    def __init__(self, ctxs, group=None):
        import os
        self.ctxs = list(ctxs)
        c0 = self.ctxs[0]
        self.npz, self.bd, self.grid, self.lib = c0.npz, c0.bd, c0.grid, c0.lib
        if group is None:
            group = os.environ.get("FV3_MI355X_FACE_GROUP", "1") != "0"
        self.group = FaceGroup(self.ctxs) if (group and len(self.ctxs) > 1) else None
        self.stream = c0.stream

    def flush(self):
        if self.group:
            self.group.flush()

    def zeros(self, kind, nk=None):
        return FaceSet([c.zeros(kind, nk) for c in self.ctxs])

    def empty(self, kind, nk=None):
        return FaceSet([c.empty(kind, nk) for c in self.ctxs])

    def from_host(self, a):
        if isinstance(a, (list, tuple)):
            return FaceSet([c.from_host(x) for c, x in zip(self.ctxs, a)])
        return FaceSet([c.from_host(a) for c in self.ctxs])

    def sync(self):
        for c in self.ctxs:
            c.sync()

    def close(self):
        if self.group:
            self.group.close()
            self.group = None
        for c in self.ctxs:
            c.close()

    def __getattr__(self, name):
        fns = [getattr(c, name) for c in self.__dict__["ctxs"]]

        def call(*args, **kw):
            out = []
            for t, fn in enumerate(fns):
                a = [x[t] if isinstance(x, FaceSet) else x for x in args]
                k = {n: (x[t] if isinstance(x, FaceSet) else x) for n, x in kw.items()}
                out.append(fn(*a, **k))
            return out if any(o is not None for o in out) else None
        return call


class CubeHaloAdapter:
    """group halo updates of the single-domain host code -> CubeHalo gathers"""
    overlaps = False
    overlaps_groups = False
    world = 1

    def __init__(self, mctx: MultiContext, npx: int, topo=None):
        self.cube = CubeHalo(mctx.ctxs, npx, topo=topo)

    def update(self, fields):
        fields = list(fields)
        kinds = [k for _, k in fields]
        if kinds == ["U", "V"]:
            self.cube.update("D", (fields[0][0].a, fields[1][0].a))
        elif kinds == ["V", "U"]:
            self.cube.update("C", (fields[0][0].a, fields[1][0].a))
        else:
            for f, k in fields:
                if k not in ("A", "B"):
                    raise ValueError(f"no cubed-sphere halo update for a lone field of kind {k}")
                self.cube.update(k, f.a)

    def start(self, fields, defer=False):
        self.update(fields)
        return None

    def post(self, pending):
        pass

    def finish(self, pending):
        pass

    def sync_edges(self, u, v):
        """mpp_get_boundary(u, v) of the last substep (dyn_core.F90:1151-1163)"""
        self.cube.update("Dedge", (u.a, v.a))


class CubeRankAdapter:
    """one face per rank: group halo updates of the single-domain host code -> CubeHaloRank messages (RCCL / gloo).  Scalar
    fields of one group with the same kind and level count travel in ONE message per neighbouring face, like the
    reference's complete=.false./.true. grouping (dyn_core.F90:823-824)."""
    overlaps = False          # no interior / rest split of d_sw on a face (the frame kernels own the edges)
    overlaps_groups = True    # start() ... finish() pairs of the substep loop keep their messages in flight across kernels

    def __init__(self, ctx, face: int, npx: int, dist, topo=None):
        self.cube = CubeHaloRank(ctx, face, npx, dist, topo=topo)
        self.world = 6
        self.rank = face

    def _groups(self, fields):
        fields = list(fields)
        kinds = [k for _, k in fields]
        if kinds == ["U", "V"]:
            return [("D", [fields[0][0], fields[1][0]])]
        if kinds == ["V", "U"]:
            return [("C", [fields[0][0], fields[1][0]])]
        groups = {}
        for f, k in fields:
            if k not in ("A", "B"):
                raise ValueError(f"no cubed-sphere halo update for a lone field of kind {k}")
            groups.setdefault((k, tuple(f.shape[2:])), []).append(f)
        return [(k, fs) for (k, _), fs in groups.items()]

    def start(self, fields, defer=False):
        return [self.cube.start(k, fs) for k, fs in self._groups(fields)]

    def post(self, pending):
        pass

    def finish(self, pending):
        for h in pending or ():
            self.cube.finish(h)

    def update(self, fields):
        self.finish(self.start(fields))

    def sync_edges(self, u, v):
        """mpp_get_boundary(u, v) of the last substep (dyn_core.F90:1151-1163)"""
        self.cube.update("Dedge", [u, v])


class CubeNativeAdapter:
    """group halo updates of the single-domain host code -> the cube-edge exchange behind the C ABI (cubed_halo.CubeHaloNative).
    ctx: a MultiContext holding the faces of this rank (all six, or one), or a single Context of one face.  Every field of a group
    travels in ONE message per neighbouring face (the reference's complete=.false./.true. grouping, dyn_core.F90:823-824)."""
    overlaps = False
    overlaps_groups = True    # start() ... finish() pairs keep their messages in flight across kernels (one group at a time)

    def __init__(self, ctx, faces, face_rank, rank: int = 0, nranks: int = 1, unique_id: bytes | None = None):
        self.multi = hasattr(ctx, "ctxs")
        ctxs = ctx.ctxs if self.multi else [ctx]
        self.cube = CubeHaloNative(ctxs, faces, face_rank, rank, nranks, unique_id)
        self.world, self.rank = nranks, rank

    def _members(self, f):
        return list(f.a) if self.multi else [f]

    def _groups(self, fields):
        fields = list(fields)
        kinds = [k for _, k in fields]
        if kinds == ["U", "V"]:
            return [("D", (self._members(fields[0][0]), self._members(fields[1][0])))]
        if kinds == ["V", "U"]:
            return [("C", (self._members(fields[0][0]), self._members(fields[1][0])))]
        out = []
        for f, k in fields:
            if k not in ("A", "B"):
                raise ValueError(f"no cubed-sphere halo update for a lone field of kind {k}")
            out.append((k, self._members(f)))
        return out

    def start(self, fields, defer=False):
        self.cube.start(self._groups(fields))
        return True

    def post(self, pending):
        pass

    def finish(self, pending):
        if pending:
            self.cube.finish()

    def update(self, fields):
        self.cube.start(self._groups(fields))
        self.cube.finish()

    def sync_edges(self, u, v):
        """mpp_get_boundary(u, v) of the last substep (dyn_core.F90:1151-1163)"""
        self.cube.update("Dedge", (self._members(u), self._members(v)))


def make_sphere_contexts(npx: int, npz: int, lib=None, sphere: CubedSphere | None = None):
    cs = sphere or CubedSphere(npx)
    gs = [cs.gridstruct(t) for t in range(6)]
    ctxs = [Context(g, npz, lib=lib) for g in gs]
    return cs, gs, MultiContext(ctxs)
