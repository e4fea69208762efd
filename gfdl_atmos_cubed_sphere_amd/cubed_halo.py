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
        # faces on streams of their own (their kernels overlap on the GPU): a gather reads and writes all six faces, so the
        # gather stream (face 1's) first waits for the other five and they wait for the gather afterwards
        handles = [getattr(c, "stream", 0) for c in self.ctxs]
        self._streams = None
        if len(set(handles)) > 1 and not getattr(self.lib, "host_memory", False):
            import torch
            self._streams = [torch.cuda.ExternalStream(h) if h else torch.cuda.default_stream() for h in handles]
            self._events = [torch.cuda.Event() for _ in handles]

    def _fan_in(self):
        if self._streams:
            for t in range(1, 6):
                self._events[t].record(self._streams[t])
                self._streams[0].wait_event(self._events[t])

    def _fan_out(self):
        if self._streams:
            self._events[0].record(self._streams[0])
            for t in range(1, 6):
                self._streams[t].wait_event(self._events[0])

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
        handle = self._handle(kind, vector)
        self._fan_in()
        self.lib.check(self.lib.dll.fv3_gather_run(self.ctx.h, handle, C.c_int(nk), C.c_int(n), parr, sarr), "fv3_gather_run")
        self._fan_out()

    def close(self):
        for h in self._handles.values():
            self.lib.dll.fv3_gather_destroy(h)
        self._handles = {}


class CubeHaloRank:
    """One face per rank (BASELINE configs[4]: six MI355X, one face each): the same topology tables split by source face.

    For every peer face the entries of its halo that read THIS face are a pack list (values gathered, with the sign of the
    vector rotation applied, into one message buffer per peer) and the entries of this face's halo that read the peer are
    an unpack list.  One gather launch packs all messages of a field group, one grouped send/recv (torch.distributed:
    RCCL on GPUs -- every pair of faces that share an edge is one xGMI hop -- gloo on the host-emulation build), one gather
    launch unpacks.  Sender and receiver walk the table of the receiving face in the same order, so a message needs no
    header.  Reference: mpp_update_domains / mpp_get_boundary on the cubed-sphere mosaic, tools/fv_mp_mod.F90:498-546."""

    SLOT0 = 8          # pointer slots 0..7: the fields of a group; 8..11: the message buffers of up to four peers

    def __init__(self, ctx, face: int, npx: int, dist, ng: int = 3, topo: CubeTopology | None = None):
        self.ctx, self.face, self.dist = ctx, face, dist
        self.lib = ctx.lib
        self.topo = topo or CubeTopology(npx, ng)
        self._plans = {}
        self._bufs = {}
        self._views = {}

    # ---- tables -----------------------------------------------------------------------------------------------------------
    def _plan(self, kind: str, vector: bool, nf: int):
        """nf: number of scalar fields of the group (kinds 'A' / 'B'); vector pairs have the two members as fields 0, 1"""
        key = (kind, vector, nf)
        if key in self._plans:
            return self._plans[key]
        tab = self.topo.boundary_table() if kind == "Dedge" else self.topo.table(kind)
        pair = kind in ("D", "C", "Dedge")
        r = self.face
        peers = sorted({int(t2) for m in range(len(tab[r])) for t2 in np.unique(tab[r][m]["tile"])} |
                       {t for t in range(6) if t != r and any((tb["tile"] == r).any() for tb in tab[t])})
        assert r not in peers and len(peers) <= 4, peers
        pack = [[], [], [], [], []]
        unpack = [[], [], [], [], []]
        n_send, n_recv = [], []
        for pi, s in enumerate(peers):
            pos = 0
            for f in range(nf if not pair else 2):                 # what face s needs from me, in the order of ITS table
                tb = tab[s][f if pair else 0]
                sel = tb["tile"] == r
                n = int(sel.sum())
                src_f = np.where(tb["comp"][sel] == 0, f, 1 - f) if pair else np.full(n, f)
                pack[0].append(np.full(n, self.SLOT0 + pi)); pack[1].append(pos + np.arange(n))
                pack[2].append(src_f); pack[3].append(tb["src"][sel])
                pack[4].append(tb["sign"][sel] if (vector and pair) else np.ones(n, dtype=np.int64))
                pos += n
            n_send.append(pos)
            pos = 0
            for f in range(nf if not pair else 2):                 # what I need from face s, in the order of MY table
                tb = tab[r][f if pair else 0]
                sel = tb["tile"] == s
                n = int(sel.sum())
                unpack[0].append(np.full(n, f)); unpack[1].append(tb["dst"][sel])
                unpack[2].append(np.full(n, self.SLOT0 + pi)); unpack[3].append(pos + np.arange(n))
                unpack[4].append(np.ones(n, dtype=np.int64))
                pos += n
            n_recv.append(pos)
        ip = C.POINTER(C.c_int)

        def make(cols):
            arrs = [np.ascontiguousarray(np.concatenate(c), dtype=np.int32) for c in cols]
            h = _vp()
            self.lib.check(self.lib.dll.fv3_gather_create(self.ctx.h, C.c_int(arrs[0].size), *[a.ctypes.data_as(ip) for a in arrs],
                                                         C.byref(h)), "fv3_gather_create")
            return h
        plan = dict(peers=peers, n_send=n_send, n_recv=n_recv, pack=make(pack), unpack=make(unpack))
        self._plans[key] = plan
        return plan

    def _tensor(self, dev):
        import torch
        if dev.ptr not in self._views:
            if getattr(self.lib, "host_memory", False):
                n = int(np.prod(dev.shape))
                flat = np.ctypeslib.as_array(C.cast(C.c_void_p(dev.ptr), C.POINTER(C.c_double)), (n,))
                self._views[dev.ptr] = torch.from_numpy(flat)
            else:
                self._views[dev.ptr] = torch.as_tensor(dev, device="cuda")
        return self._views[dev.ptr]

    def _buffers(self, key, plan, nk):
        from .lib import DeviceArray
        bk = key + (nk,)
        if bk not in self._bufs:
            send = [DeviceArray(self.ctx, (n * nk,)) for n in plan["n_send"]]
            recv = [DeviceArray(self.ctx, (n * nk,)) for n in plan["n_recv"]]
            self._bufs[bk] = (send, recv, [self._tensor(b) for b in send], [self._tensor(b) for b in recv])
        return self._bufs[bk]

    def _run(self, handle, nk, fields, bufs, lens):
        ptrs = [f.ptr for f in fields] + [0] * (self.SLOT0 - len(fields)) + [b.ptr for b in bufs]
        strides = [int(f.shape[0] * f.shape[1]) for f in fields] + [0] * (self.SLOT0 - len(fields)) + [int(n) for n in lens]
        n = len(ptrs)
        self.lib.check(self.lib.dll.fv3_gather_run(self.ctx.h, handle, C.c_int(nk), C.c_int(n), (C.c_void_p * n)(*ptrs),
                                                   (C.c_size_t * n)(*strides)), "fv3_gather_run")

    # ---- the update -------------------------------------------------------------------------------------------------------
    def start(self, kind: str, fields, vector: bool = True):
        """fields: the DeviceArrays of one group -- same kind and level count ('A' / 'B': up to 8 scalars; 'D' / 'C' / 'Dedge':
        the two members of the pair).  Packs and posts the messages; finish() waits and unpacks."""
        fields = list(fields)
        pair = kind in ("D", "C", "Dedge")
        assert (len(fields) == 2) if pair else (1 <= len(fields) <= self.SLOT0)
        shp = fields[0].shape
        nk = 1 if len(shp) == 2 else int(np.prod(shp[2:]))
        key = (kind, vector, len(fields))
        plan = self._plan(*key)
        send, recv, tsend, trecv = self._buffers(key, plan, nk)
        self._run(plan["pack"], nk, fields, send, plan["n_send"])
        dist = self.dist
        host = getattr(self.lib, "host_memory", False)
        p2p = []
        for pi, s in enumerate(plan["peers"]):          # every pair of ranks posts in ascending peer order on both sides
            p2p.append(dist.P2POp(dist.isend, tsend[pi], s))
            p2p.append(dist.P2POp(dist.irecv, trecv[pi], s))
        if host:
            works = dist.batch_isend_irecv(p2p)
        else:
            import torch
            st = getattr(self.ctx, "stream", 0)
            stream = torch.cuda.ExternalStream(st) if st else torch.cuda.default_stream()
            with torch.cuda.stream(stream):              # RCCL orders its stream after the pack kernel's
                works = dist.batch_isend_irecv(p2p)       # ncclGroupStart ... ncclGroupEnd
        return dict(plan=plan, nk=nk, fields=fields, recv=recv, works=works, host=host)

    def finish(self, h):
        if h["host"]:
            for w in h["works"]:
                w.wait()
        else:
            import torch
            st = getattr(self.ctx, "stream", 0)
            stream = torch.cuda.ExternalStream(st) if st else torch.cuda.default_stream()
            with torch.cuda.stream(stream):
                for w in h["works"]:
                    w.wait()
        self._run(h["plan"]["unpack"], h["nk"], h["fields"], h["recv"], h["plan"]["n_recv"])

    def update(self, kind: str, fields, vector: bool = True):
        self.finish(self.start(kind, fields, vector))

    def close(self):
        for p in self._plans.values():
            self.lib.dll.fv3_gather_destroy(p["pack"])
            self.lib.dll.fv3_gather_destroy(p["unpack"])
        self._plans = {}


class CubeHaloNative:
    """The cube-edge exchange behind the C ABI (fv3_cube_halo_start / _complete: pack kernel, grouped RCCL sends / receives on the
    communication stream of the first context, unpack kernel; the library's own topology, csrc/cube_topo.h).  `ctxs` are the faces
    this rank holds -- all six on one GPU (every message then travels through RCCL to the same rank: the loopback of the message
    path) or one per rank (BASELINE config 5: six MI355X).  `unique_id`: the 128 bytes of rank 0's fv3_comm_get_unique_id,
    distributed by the caller."""

    def __init__(self, ctxs, faces, face_rank, rank: int = 0, nranks: int = 1, unique_id: bytes | None = None):
        self.ctxs, self.faces, self.face_rank = list(ctxs), [int(f) for f in faces], [int(r) for r in face_rank]
        assert self.faces == sorted(self.faces) and len(self.ctxs) == len(self.faces)
        self.unique_id = self.ctxs[0].comm_init(rank, nranks, unique_id)
        self._pending = False

    def start(self, groups):
        """groups: list of (kind, members[, scalar_pair]); members = list of per-context arrays ('A' / 'B') or a pair of such lists"""
        from .lib import cube_halo_start
        if self._pending:
            raise RuntimeError("CubeHaloNative: one group in flight per context (complete the previous one first)")
        norm = []
        for g in groups:
            kind, mem = g[0], g[1]
            sp = g[2] if len(g) > 2 else False
            if kind in ("D", "C", "Dedge"):
                norm.append((kind, list(mem[0]), list(mem[1]), sp))
            else:
                norm.append((kind, list(mem), None, sp))
        for n in range(0, len(norm), 8):
            if n:
                self.finish()
            cube_halo_start(self.ctxs, self.faces, self.face_rank, norm[n:n + 8])
            self._pending = True

    def finish(self):
        from .lib import cube_halo_complete
        if self._pending:
            try:
                cube_halo_complete(self.ctxs)
            finally:
                self._pending = False      # also when the library refuses: it drops its pending group itself

    def update(self, kind: str, fields, vector: bool = True):
        """the interface of CubeHalo.update: 'A' / 'B': list of per-context arrays; pairs: (list, list)"""
        self.start([(kind, fields, not vector)])
        self.finish()

    def close(self):
        pass
