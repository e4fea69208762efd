"""ctypes binding of the product library libfv3_mi355x.so (C ABI: include/fv3_mi355x.h).

There is NO CPU fallback: ``load()`` raises if the HIP library has not been built
(``python -c "import __graft_entry__ as g; g.build()"``), and every entry point raises
``Fv3Error`` on a non-zero status.  Device memory comes from the library's own ``fv3_malloc``
(hipMalloc); ``DeviceArray`` exposes ``__cuda_array_interface__`` so that torch (used only for
streams and torch.distributed) can alias a buffer without copying.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .grid import GridStruct
from .layout import Bounds

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
PRODUCT_SO = os.path.join(_CSRC, "libfv3_mi355x.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_vp = C.c_void_p

_A = ["area", "rarea", "dxa", "dya", "rdxa", "rdya", "cosa_s", "rsin2", "f0"]
_U = ["dx", "rdx", "dyc", "rdyc", "cosa_v", "sina_v", "rsin_v", "divg_u", "del6_u"]
_V = ["dy", "rdy", "dxc", "rdxc", "cosa_u", "sina_u", "rsin_u", "divg_v", "del6_v"]
_B = ["rarea_c", "fC", "cosa", "sina"]

# every symbol include/fv3_mi355x.h declares (tests check the built library exports all of them)
EXPORTS = ["fv3_last_error", "fv3_create", "fv3_destroy", "fv3_set_stream", "fv3_group_create", "fv3_group_flush", "fv3_group_stats", "fv3_group_destroy", "fv3_grid_upload", "fv3_grid_upload_cubed", "fv3_gather_create", "fv3_gather_run", "fv3_gather_destroy", "fv3_grid_geom", "fv3_malloc",
           "fv3_free", "fv3_memcpy_h2d", "fv3_memcpy_d2h", "fv3_memcpy_d2d", "fv3_memset", "fv3_sync", "fv3_registry_mode", "fv3_registry_put", "fv3_registry_get", "fv3_registry_host_touched", "fv3_registry_fetch", "fv3_registry_forget", "fv3_registry_stats", "fv3_fv_tp_2d", "fv3_ppm_line", "fv3_c_sw",
           "fv3_dsw_levels_upload", "fv3_d_sw", "fv3_d_sw_interior", "fv3_d_sw_rest", "fv3_halo_fill_periodic", "fv3_halo_message_elems", "fv3_halo_periodic_group", "fv3_halo_pack",
           "fv3_halo_unpack", "fv3_pt_to_theta_v", "fv3_c2l", "fv3_rayleigh_u2f", "fv3_rayleigh_apply", "fv3_rayleigh_super", "fv3_compute_total_energy", "fv3_energy_fixer_sums", "fv3_remap_finish", "fv3_ordered_sum", "fv3_adv_pe", "fv3_omga_update", "fv3_divg2_ext", "fv3_one_grad_p", "fv3_one_grad_p_nh", "fv3_copy_a_to_cc", "fv3_heat_source_accum", "fv3_del2_cubed", "fv3_apply_heat_source", "fv3_profile", "fv3_profile_report", "fv3_comm_get_unique_id", "fv3_comm_init", "fv3_comm_destroy", "fv3_halo_start", "fv3_halo_complete", "fv3_allreduce_max", "fv3_cube_table", "fv3_cube_halo_start", "fv3_cube_halo_complete",
           "fv3_set_dp_ref", "fv3_update_dz_c", "fv3_set_condensate", "fv3_set_fast_tau_w", "fv3_set_ray_fast", "fv3_ray_fast", "fv3_mix_dp", "fv3_compute_aam", "fv3_consv_am_apply", "fv3_riem_solver_c", "fv3_update_dz_d", "fv3_riem_solver3",
           "fv3_p_grad_c", "fv3_nh_p_grad", "fv3_split_p_grad", "fv3_grad1_p_update", "fv3_d_sw_inline_q", "fv3_flux_accum", "fv3_fill2d_mass", "fv3_fill2d_apply", "fv3_set_remap_te", "fv3_profile_report_timers", "fv3_prt_maxmin", "fv3_pk3_halo", "fv3_pe_halo", "fv3_geopk", "fv3_zh_from_delz", "fv3_set_ak_bk", "fv3_set_moist", "fv3_lagrangian_to_eulerian",
           "fv3_tracer_2d_prep", "fv3_tracer_2d_scale", "fv3_tracer_2d_step"]


class Fv3Error(RuntimeError):
    pass


class _Domain(C.Structure):
    _fields_ = [(n, C.c_int) for n in ["is_", "ie", "js", "je", "ng", "npx", "npy", "npz", "grid_type", "do_diss_est",
                                       "prevent_diss_cooling", "stretched_grid"]] + [("lim_fac", C.c_double)]


class _GridHost(C.Structure):
    _fields_ = [("da_min", C.c_double), ("da_min_c", C.c_double)] + [(n, _dp) for n in
                                                                      _A + _U + _V + _B + ["sin_sg", "cos_sg"]]


class _GridCubed(C.Structure):
    _fields_ = [(n, _dp) for n in ["edge_w", "edge_e", "edge_s", "edge_n", "rsina"]] + [("corner_f", C.c_double * 12)] + [
        (n, _dp) for n in ["a11", "a12", "a21", "a22", "ec1", "ec2", "en1", "en2"]]


class _CubeField(C.Structure):
    _fields_ = [("kind", C.c_int), ("f0", C.c_void_p), ("f1", C.c_void_p), ("nk", C.c_int), ("scalar_pair", C.c_int)]


CUBE_KINDS = {"A": 0, "B": 1, "D": 2, "C": 3, "Dedge": 4}


def cube_table(lib, npx: int, kind: str, member: int, face: int, ng: int = 3):
    """fv3_cube_table: the library's own halo topology of the cubed sphere (csrc/cube_topo.h) -> dict(dst, tile, comp, src, sign)"""
    dll = lib.dll
    dll.fv3_cube_table.restype = C.c_long
    args = (C.c_int(npx), C.c_int(ng), C.c_int(CUBE_KINDS[kind]), C.c_int(member), C.c_int(face))
    n = dll.fv3_cube_table(*args, None, None, None, None, None)
    if n < 0:
        raise Fv3Error("fv3_cube_table: bad argument")
    dst, src = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64)
    st, cp, sg = (np.zeros(n, dtype=np.int32) for _ in range(3))
    lp, ip = C.POINTER(C.c_long), C.POINTER(C.c_int)
    dll.fv3_cube_table(*args, dst.ctypes.data_as(lp), st.ctypes.data_as(ip), cp.ctypes.data_as(ip), src.ctypes.data_as(lp), sg.ctypes.data_as(ip))
    return dict(dst=dst, tile=st.astype(np.int64), comp=cp.astype(np.int64), src=src, sign=sg.astype(np.int64))


def cube_halo_start(ctxs, faces, face_rank, groups):
    """fv3_cube_halo_start.  ctxs: the contexts of the faces this rank holds (faces ascending; ctxs[0] has the communicator);
    groups: list of (kind, f0s, f1s, scalar_pair) with f0s / f1s = one DeviceArray per context (f1s None for kinds 'A' / 'B')"""
    lib = ctxs[0].lib
    n, nf = len(ctxs), len(groups)
    arr = (_CubeField * (n * nf))()
    for i in range(n):
        for f, (kind, f0s, f1s, sp) in enumerate(groups):
            a0 = f0s[i]
            e = arr[i * nf + f]
            e.kind, e.f0, e.f1 = CUBE_KINDS[kind], a0.ptr, (f1s[i].ptr if f1s is not None else None)
            e.nk = int(np.prod(a0.shape[2:])) if len(a0.shape) > 2 else 1
            e.scalar_pair = int(bool(sp))
    hs = (C.c_void_p * n)(*[c.h for c in ctxs])
    lib.check(lib.dll.fv3_cube_halo_start(C.c_int(n), hs, (C.c_int * n)(*faces), (C.c_int * 6)(*face_rank), C.c_int(nf), arr),
              "fv3_cube_halo_start")


def cube_halo_complete(ctxs):
    lib = ctxs[0].lib
    n = len(ctxs)
    lib.check(lib.dll.fv3_cube_halo_complete(C.c_int(n), (C.c_void_p * n)(*[c.h for c in ctxs])), "fv3_cube_halo_complete")


class _DswParams(C.Structure):
    _fields_ = [("dt", C.c_double)] + [(n, C.c_int) for n in ["hord_tr", "hord_mt", "hord_vt", "hord_tm", "hord_dp"]] + [
        (n, C.c_double) for n in ["dddmp", "d4_bg", "kgb"]] + [("hydrostatic", C.c_int), ("use_cond", C.c_int)]


class _DswLevels(C.Structure):
    _fields_ = [(n, _ip) for n in ["nord_k", "nord_v", "nord_w", "nord_t"]] + [(n, _dp) for n in
                                                                                 ["d2_divg", "damp_vt", "damp_w",
                                                                                  "damp_t", "d_con_k"]]


class _NhConsts(C.Structure):
    _fields_ = [(n, C.c_double) for n in ["grav", "rdgas", "cp_air", "akap", "ptop", "p_fac", "a_imp"]] + [("m_split", C.c_int)]


# FMS constants_mod (GFDL defaults, FMS 2024.03 constants/gfdl_constants.fh): not in the reference tree
GRAV, RDGAS, KAPPA = 9.80, 287.04, 2.0 / 7.0
CP_AIR = RDGAS / KAPPA


def nh_consts(ptop, p_fac=0.05, a_imp=1.0, akap=KAPPA, grav=GRAV, rdgas=RDGAS, cp_air=CP_AIR, m_split=1):
    return dict(grav=grav, rdgas=rdgas, cp_air=cp_air, akap=akap, ptop=ptop, p_fac=p_fac, a_imp=a_imp, m_split=m_split)


class _RemapParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in ["last_step", "hydrostatic", "adiabatic", "nq", "kord_mt", "kord_wz", "kord_tm",
                                       "sphum"]] + [(n, C.c_double) for n in ["akap", "ptop", "rdgas", "grav", "cv_air",
                                                                              "r_vir", "cp", "t_min"]] + [
        ("fill", C.c_int)]


class _MoistParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in ["moist_kappa", "use_cond", "nwat", "sphum", "liq_wat", "rainwat", "ice_wat",
                                       "snowwat", "graupel"]] + [(n, C.c_double) for n in ["cv_vap", "c_liq", "c_ice"]]


class Fv3Lib:
    """A loaded shared object exporting the fv3_* C ABI."""

    def __init__(self, path: str):
        if not os.path.exists(path):
            raise Fv3Error(f"{path} not found: the HIP library is not built (run __graft_entry__.build()); "
                           "there is no CPU fallback")
        self.path = path
        self.host_memory = "hostemu" in os.path.basename(path)   # the tests' logic harness, never the product
        if not self.host_memory:
            # torch (streams, torch.distributed for the halo exchange) ships its own libamdhip64 with the same SONAME as
            # the one this library is linked to; whichever is mapped first serves the whole process.  Map torch's first:
            # the other order (this library's runtime initialised, torch importing later) leaves torch without a device
            # ("No HIP GPUs are available").
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        self.dll = C.CDLL(path)
        missing = [s for s in EXPORTS if not hasattr(self.dll, s)]
        if missing:
            raise Fv3Error(f"{path} does not export {missing}")
        self.dll.fv3_last_error.restype = C.c_char_p

    def check(self, rc: int, what: str):
        if rc != 0:
            raise Fv3Error(f"{what}: {self.dll.fv3_last_error().decode()}")


_PRODUCT: Fv3Lib | None = None


def build_id() -> str:
    """Short hash of the kernel / C-ABI sources the product library is built from: measurements kept under profiles/
    (HBM traffic counters) carry it so that bench.py can tell a stale file from one taken on the build it is timing."""
    import hashlib
    h = hashlib.sha256()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = [os.path.join(_CSRC, f) for f in sorted(os.listdir(_CSRC)) if f.endswith((".h", ".hip"))]
    files += [os.path.join(root, "include", f) for f in sorted(os.listdir(os.path.join(root, "include"))) if f.endswith(".h")]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


def load() -> Fv3Lib:
    """The product library (HIP, gfx950).  Raises if it is missing."""
    global _PRODUCT
    if _PRODUCT is None:
        # FV3_MI355X_SO selects another build of the same HIP library (e.g. a contraction-on build)
        _PRODUCT = Fv3Lib(os.environ.get("FV3_MI355X_SO", PRODUCT_SO))
    return _PRODUCT


class HaloField(C.Structure):
    _fields_ = [("field", C.POINTER(C.c_double)), ("kind", C.c_int), ("nk", C.c_int)]


class DeviceArray:
    """A device buffer holding one field in the reference layout (Fortran order)."""

    def __init__(self, ctx: "Context", shape):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in shape)
        self.nbytes = int(np.prod(self.shape)) * 8
        ptr = _vp()
        ctx.lib.check(ctx.lib.dll.fv3_malloc(C.byref(ptr), C.c_size_t(self.nbytes)), "fv3_malloc")
        self.ptr = ptr.value
        ctx._buffers.append(self)

    @property
    def p(self):
        return C.cast(_vp(self.ptr), _dp)

    @property
    def __cuda_array_interface__(self):
        # Fortran-ordered strides
        strides, acc = [], 8
        for s in self.shape:
            strides.append(acc)
            acc *= s
        return {"shape": self.shape, "typestr": "<f8", "data": (self.ptr, False), "version": 3,
                "strides": tuple(strides)}

    def upload(self, a: np.ndarray):
        a = np.asfortranarray(a, dtype=np.float64)
        assert a.shape == self.shape, (a.shape, self.shape)
        self.ctx.lib.check(self.ctx.lib.dll.fv3_memcpy_h2d(self.ctx.h, _vp(self.ptr), a.ctypes.data_as(_vp),
                                                           C.c_size_t(self.nbytes)), "fv3_memcpy_h2d")
        self.ctx.sync()  # pageable host memory: keep `a` alive until the copy is done
        return self

    def download(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=np.float64, order="F")
        self.ctx.lib.check(self.ctx.lib.dll.fv3_memcpy_d2h(self.ctx.h, out.ctypes.data_as(_vp), _vp(self.ptr),
                                                           C.c_size_t(self.nbytes)), "fv3_memcpy_d2h")
        self.ctx.sync()
        return out

    def copy_from(self, other: "DeviceArray"):
        assert other.nbytes == self.nbytes
        self.ctx.lib.check(self.ctx.lib.dll.fv3_memcpy_d2d(self.ctx.h, _vp(self.ptr), _vp(other.ptr),
                                                           C.c_size_t(self.nbytes)), "fv3_memcpy_d2d")
        return self

    def zero(self):
        self.ctx.lib.check(self.ctx.lib.dll.fv3_memset(self.ctx.h, _vp(self.ptr), 0, C.c_size_t(self.nbytes)),
                           "fv3_memset")
        return self

    def free(self):
        if self.ptr:
            self.ctx.lib.dll.fv3_free(_vp(self.ptr))
            self.ptr = None


def _pp(x):
    return None if x is None else x.p


class FaceGroup:
    """fv3_group: the faces (contexts) one rank holds, launched together -- every member queues its launches, and when all of them
    have issued the same kernel ONE launch runs them (include/fv3_mi355x.h).  The members share the first member's stream."""

    def __init__(self, ctxs):
        self.lib = ctxs[0].lib
        self.ctxs = list(ctxs)
        n = len(self.ctxs)
        self.h = _vp()
        self.lib.check(self.lib.dll.fv3_group_create((C.c_void_p * n)(*[c.h.value for c in self.ctxs]), C.c_int(n),
                                                     C.byref(self.h)), "fv3_group_create")
        for c in self.ctxs:
            c.stream = self.ctxs[0].stream
            c.group = self

    def flush(self):
        self.lib.check(self.lib.dll.fv3_group_flush(self.h), "fv3_group_flush")

    def stats(self):
        """(launches that ran all members at once, launches that ran alone) since the last call"""
        a, b = C.c_long(0), C.c_long(0)
        self.lib.check(self.lib.dll.fv3_group_stats(self.h, C.byref(a), C.byref(b)), "fv3_group_stats")
        return a.value, b.value

    def close(self):
        if self.h:
            self.lib.check(self.lib.dll.fv3_group_destroy(self.h), "fv3_group_destroy")
            self.h = None
            for c in self.ctxs:
                c.group = None


class Context:
    """fv3_ctx: one rank's block of the domain + its gridstruct on the device."""

    def __init__(self, grid: GridStruct, npz: int, lib: Fv3Lib | None = None, stream: int | None = None):
        self.lib = lib or load()
        self.grid = grid
        self.bd: Bounds = grid.bd
        self.npz = npz
        self._buffers: list[DeviceArray] = []
        self.group = None      # lib.FaceGroup this context is a member of
        d = _Domain()
        b = grid.bd
        d.is_, d.ie, d.js, d.je, d.ng = b.is_, b.ie, b.js, b.je, b.ng
        d.npx, d.npy, d.npz, d.grid_type = grid.npx, grid.npy, npz, grid.grid_type
        d.do_diss_est, d.prevent_diss_cooling = int(grid.do_diss_est), int(grid.prevent_diss_cooling)
        d.stretched_grid, d.lim_fac = int(grid.stretched_grid), grid.lim_fac
        self.h = _vp()
        self.lib.check(self.lib.dll.fv3_create(C.byref(d), C.byref(self.h)), "fv3_create")
        self.stream = 0
        if stream is not None:
            self.set_stream(stream)
        gh = _GridHost()
        gh.da_min, gh.da_min_c = grid.da_min, grid.da_min_c
        keep = []
        for n in _A + _U + _V + _B + ["sin_sg", "cos_sg"]:
            a = np.asfortranarray(grid.m[n], dtype=np.float64)
            keep.append(a)
            setattr(gh, n, a.ctypes.data_as(_dp))
        self.lib.check(self.lib.dll.fv3_grid_upload(self.h, C.byref(gh)), "fv3_grid_upload")
        self.geom = int(self.lib.dll.fv3_grid_geom(self.h))  # 0 general, 1 orthogonal, 2 orthogonal + uniform
        if grid.grid_type < 3:      # a face of the cubed sphere: edge weights, rsina, corner extrapolation factors
            gc = _GridCubed()
            for n in ("edge_w", "edge_e", "edge_s", "edge_n", "rsina") + (("a11", "a12", "a21", "a22") if "a11" in grid.m else ()) + \
                    (("ec1", "ec2", "en1", "en2") if "en1" in grid.m else ()):
                a = np.asfortranarray(grid.m[n], dtype=np.float64)
                keep.append(a)
                setattr(gc, n, a.ctypes.data_as(_dp))
            for k, v in enumerate(np.asarray(grid.m["corner_f"], dtype=np.float64).ravel()):
                gc.corner_f[k] = v
            self.lib.check(self.lib.dll.fv3_grid_upload_cubed(self.h, C.byref(gc)), "fv3_grid_upload_cubed")

    # -- plumbing ------------------------------------------------------------------------------
    def set_stream(self, stream: int):
        self.stream = int(stream or 0)      # raw hipStream_t of every launch of this context (0 = the null stream)
        self.lib.check(self.lib.dll.fv3_set_stream(self.h, _vp(stream)), "fv3_set_stream")

    def sync(self):
        self.lib.check(self.lib.dll.fv3_sync(self.h), "fv3_sync")

    def empty(self, kind: str, nk: int | None = None) -> DeviceArray:
        return DeviceArray(self, self.bd.shape(kind, nk))

    def zeros(self, kind: str, nk: int | None = None) -> DeviceArray:
        return self.empty(kind, nk).zero()

    def from_host(self, a: np.ndarray) -> DeviceArray:
        return DeviceArray(self, a.shape).upload(a)

    def close(self):
        for b in self._buffers:
            b.free()
        self._buffers.clear()
        if self.h:
            self.lib.dll.fv3_destroy(self.h)
            self.h = None

    # -- operators (argument names follow the reference routines) ------------------------------------
    def fv_tp_2d(self, q, crx, cry, hord, fx, fy, xfx, yfx, ra_x=None, ra_y=None, mfx=None, mfy=None, mass=None,
                 nord=-1, damp_c=0.0, nk=None):
        """model/tp_core.F90:85 fv_tp_2d for nk slabs."""
        nk = self.npz if nk is None else nk
        self.lib.check(self.lib.dll.fv3_fv_tp_2d(self.h, C.c_int(nk), q.p, crx.p, cry.p, C.c_int(hord), fx.p, fy.p,
                                                 xfx.p, yfx.p, _pp(ra_x), _pp(ra_y), _pp(mfx), _pp(mfy), _pp(mass),
                                                 C.c_int(nord), C.c_double(damp_c)), "fv3_fv_tp_2d")

    def ppm_line(self, iord, which, h, c, flux, n):
        """xppm / yppm (model/tp_core.F90:324-1152) on one line: h = the line with 3 halo cells either side (device, n + 6), c and flux
        n + 1; which: 0 the tile kernels' operator, 1 / 2 the marching kernels' (along the lanes / register window).  Unit tests only."""
        self.lib.check(self.lib.dll.fv3_ppm_line(self.h, C.c_int(iord), C.c_int(which), h.p, c.p, flux.p, C.c_int(n)), "fv3_ppm_line")

    def c_sw(self, delpc, delp, ptc, pt, u, v, w, uc, vc, ua, va, wc, ut, vt, divg_d, nord, dt2, hydrostatic,
             dord4=True):
        """model/sw_core.F90:79 c_sw over all levels (the k loop of dyn_core.F90:436-447)."""
        self.lib.check(self.lib.dll.fv3_c_sw(self.h, delpc.p, delp.p, ptc.p, pt.p, u.p, v.p, _pp(w), uc.p, vc.p, ua.p,
                                             va.p, _pp(wc), ut.p, vt.p, divg_d.p, C.c_int(nord), C.c_double(dt2),
                                             C.c_int(int(hydrostatic)), C.c_int(int(dord4))), "fv3_c_sw")

    def dsw_levels(self, lev: dict):
        lv = _DswLevels()
        keep = []
        for n in ["nord_k", "nord_v", "nord_w", "nord_t"]:
            a = np.ascontiguousarray(lev[n], dtype=np.int32)
            assert a.size == self.npz
            keep.append(a)
            setattr(lv, n, a.ctypes.data_as(_ip))
        for n in ["d2_divg", "damp_vt", "damp_w", "damp_t", "d_con_k"]:
            a = np.ascontiguousarray(lev[n], dtype=np.float64)
            assert a.size == self.npz
            keep.append(a)
            setattr(lv, n, a.ctypes.data_as(_dp))
        self.lib.check(self.lib.dll.fv3_dsw_levels_upload(self.h, C.byref(lv)), "fv3_dsw_levels_upload")

    def d_sw(self, par: dict, delpc, delp, pt, u, v, w, uc, vc, ua, va, divg_d, mfx, mfy, cx, cy, crx, cry, xfx, yfx,
             q_con, delp_out, pt_out, u_out, v_out, w_out, q_con_out, heat_s, diss_e, phase: str = "all"):
        """model/sw_core.F90:494 d_sw over all levels (the k loop of dyn_core.F90:658-812).
        phase: "all", or "interior" (the part that needs no halo of uc, vc, divg_d) followed by "rest"."""
        fn = {"all": "fv3_d_sw", "interior": "fv3_d_sw_interior", "rest": "fv3_d_sw_rest"}[phase]
        pr = _DswParams()
        for k in ["dt", "hord_tr", "hord_mt", "hord_vt", "hord_tm", "hord_dp", "dddmp", "d4_bg", "kgb", "hydrostatic",
                  "use_cond"]:
            setattr(pr, k, par[k])
        self.lib.check(getattr(self.lib.dll, fn)(self.h, C.byref(pr), _pp(delpc), delp.p, pt.p, u.p, v.p, _pp(w), uc.p,
                                                 vc.p, ua.p, va.p, divg_d.p, mfx.p, mfy.p, cx.p, cy.p, crx.p, cry.p,
                                                 xfx.p, yfx.p, _pp(q_con), delp_out.p, pt_out.p, u_out.p, v_out.p,
                                                 _pp(w_out), _pp(q_con_out), _pp(heat_s), _pp(diss_e)), fn)

    def profile(self, enable: bool):
        self.lib.check(self.lib.dll.fv3_profile(self.h, C.c_int(int(enable))), "fv3_profile")

    def profile_report(self) -> dict:
        """{label: (count, total_ms)} measured with HIP events on the launch stream."""
        buf = C.create_string_buffer(1 << 16)
        self.lib.check(self.lib.dll.fv3_profile_report(self.h, buf, C.c_size_t(len(buf))), "fv3_profile_report")
        out = {}
        for line in buf.value.decode().splitlines():
            name, n, ms = line.split()
            out[name] = (int(n), float(ms))
        return out

    def profile_report_timers(self) -> dict:
        """the same events under the reference's timing_on / timing_off names: {timer: (count, total_ms)}"""
        buf = C.create_string_buffer(1 << 16)
        self.lib.check(self.lib.dll.fv3_profile_report_timers(self.h, buf, C.c_size_t(len(buf))), "fv3_profile_report_timers")
        out = {}
        for line in buf.value.decode().splitlines():
            name, n, ms = line.split()
            out[name] = (int(n), float(ms))
        return out

    def prt_maxmin(self, q, fac=1.0):
        """prt_mxm (tools/fv_diagnostics.F90:4265-4313) of an A-kind field: (max, min, area mean of the last level), times fac"""
        out = (C.c_double * 3)()
        nk = q.shape[2] if len(q.shape) > 2 else 1
        self.lib.check(self.lib.dll.fv3_prt_maxmin(self.h, q.p, C.c_int(nk), C.c_double(fac), out), "fv3_prt_maxmin")
        return float(out[0]), float(out[1]), float(out[2])

    # -- nonhydrostatic column path -------------------------------------------------------------------
    def _cn(self, cn: dict):
        s = _NhConsts()
        for k, val in cn.items():
            setattr(s, k, val)
        return s

    def set_dp_ref(self, dp0):
        a = np.ascontiguousarray(dp0, dtype=np.float64)
        assert a.size == self.npz
        self.lib.check(self.lib.dll.fv3_set_dp_ref(self.h, a.ctypes.data_as(_dp)), "fv3_set_dp_ref")

    def update_dz_c(self, dt, zs, ut, vt, gz_in, gz, ws):
        """model/nh_utils.F90:59 update_dz_c"""
        self.lib.check(self.lib.dll.fv3_update_dz_c(self.h, C.c_double(dt), zs.p, ut.p, vt.p, gz_in.p, gz.p, ws.p),
                       "fv3_update_dz_c")

    def riem_solver_c(self, dt, cn, hs, w3, pt, delp, gz, pef, ws):
        """model/nh_utils.F90:323 Riem_Solver_c"""
        s = self._cn(cn)
        self.lib.check(self.lib.dll.fv3_riem_solver_c(self.h, C.c_double(dt), C.byref(s), hs.p, w3.p, pt.p, delp.p,
                                                      gz.p, pef.p, ws.p), "fv3_riem_solver_c")

    def update_dz_d(self, hord, zs, zh_in, zh_out, crx, cry, xfx, yfx, ws, rdt):
        """model/nh_utils.F90:204 update_dz_d"""
        self.lib.check(self.lib.dll.fv3_update_dz_d(self.h, C.c_int(hord), zs.p, zh_in.p, zh_out.p, crx.p, cry.p,
                                                    xfx.p, yfx.p, ws.p, C.c_double(rdt)), "fv3_update_dz_d")

    def set_fast_tau_w(self, rff=None):
        """fast_tau_w_sec > 0: rff(1:k_rf) of nh_utils.F90:356-367 (host array; None / empty = off) for the following Riemann solver calls"""
        rff = np.ascontiguousarray(rff if rff is not None else [], dtype=np.float64)
        self.lib.check(self.lib.dll.fv3_set_fast_tau_w(self.h, C.c_int(len(rff)), rff.ctypes.data_as(_dp) if len(rff) else None), "fv3_set_fast_tau_w")

    def set_ray_fast(self, kmax, k_rf, dm, rf, dp):
        """what Ray_fast keeps from its first call (dyn_core.F90:2519-2545): rf(1:kmax), dp_ref(1:npz), k_rf, dm"""
        rf = np.ascontiguousarray(rf, dtype=np.float64)
        dp = np.ascontiguousarray(dp, dtype=np.float64)
        self.lib.check(self.lib.dll.fv3_set_ray_fast(self.h, C.c_int(int(kmax)), C.c_int(int(k_rf)), C.c_double(dm), rf.ctypes.data_as(_dp),
                                                     dp.ctypes.data_as(_dp)), "fv3_set_ray_fast")

    def ray_fast(self, u, v, w, hydrostatic):
        """Ray_fast (dyn_core.F90:2549-2597) on the compute domain; w may be None when hydrostatic"""
        self.lib.check(self.lib.dll.fv3_ray_fast(self.h, u.p, v.p, w.p if w is not None else None, C.c_int(int(hydrostatic))), "fv3_ray_fast")

    def mix_dp(self, hydrostatic, w, delp, pt):
        """mix_dp (dyn_core.F90:2119-2200; flagstruct%fill_dp, :820): thin layers borrow mass from their neighbour, pt and w mixed; in place"""
        self.lib.check(self.lib.dll.fv3_mix_dp(self.h, C.c_int(int(hydrostatic)), w.p if w is not None else None, delp.p, pt.p), "fv3_mix_dp")

    def set_condensate(self, q_con=None, cappa=None):
        """use_cond / moist_kappa arrays of the following riem_solver_c / riem_solver3 calls (None = .false.)"""
        self.lib.check(self.lib.dll.fv3_set_condensate(self.h, q_con.p if q_con is not None else None,
                                                       cappa.p if cappa is not None else None), "fv3_set_condensate")

    def riem_solver3(self, dt, cn, zs, w, delz, pt, delp, zh, pe, ppe, pk3, pk, peln, ws, use_logp=False,
                     last_call=False, fp_out=False):
        """model/nh_core.F90:47 Riem_Solver3"""
        s = self._cn(cn)
        self.lib.check(self.lib.dll.fv3_riem_solver3(self.h, C.c_double(dt), C.byref(s), zs.p, w.p, delz.p, pt.p,
                                                     delp.p, zh.p, _pp(pe), ppe.p, pk3.p, _pp(pk), _pp(peln), ws.p,
                                                     C.c_int(int(use_logp)), C.c_int(int(last_call)),
                                                     C.c_int(int(fp_out))), "fv3_riem_solver3")

    def p_grad_c(self, dt2, delpc, pkc, gz, uc, vc, hydrostatic):
        """model/dyn_core.F90:1635 p_grad_c"""
        self.lib.check(self.lib.dll.fv3_p_grad_c(self.h, C.c_double(dt2), delpc.p, pkc.p, gz.p, uc.p, vc.p,
                                                 C.c_int(int(hydrostatic))), "fv3_p_grad_c")

    def nh_p_grad(self, u, v, pp, gz, delp, pk, dt, top_value, gz_scale=1.0):
        """model/dyn_core.F90:1697 nh_p_grad (gz_scale: read gz*gz_scale, fusing gz = zh*grav of :982-989)"""
        self.lib.check(self.lib.dll.fv3_nh_p_grad(self.h, u.p, v.p, pp.p, gz.p, C.c_double(gz_scale), delp.p, pk.p,
                                                  C.c_double(dt), C.c_double(top_value)), "fv3_nh_p_grad")

    def zh_from_delz(self, zs, delz, zh):
        self.lib.check(self.lib.dll.fv3_zh_from_delz(self.h, zs.p, delz.p, zh.p), "fv3_zh_from_delz")

    def pk3_halo(self, ptop, akap, pk3, delp, use_logp=False):
        self.lib.check(self.lib.dll.fv3_pk3_halo(self.h, C.c_double(ptop), C.c_double(akap), pk3.p, delp.p,
                                                 C.c_int(int(use_logp))), "fv3_pk3_halo")

    def pe_halo(self, ptop, pe, delp):
        self.lib.check(self.lib.dll.fv3_pe_halo(self.h, C.c_double(ptop), pe.p, delp.p), "fv3_pe_halo")

    def geopk(self, ptop, akap, cp_air, pe, peln, delp, pk, gz, hs, pt, pkz, CG):
        """model/dyn_core.F90:2202 geopk"""
        self.lib.check(self.lib.dll.fv3_geopk(self.h, C.c_double(ptop), C.c_double(akap), C.c_double(cp_air),
                                              C.c_double(ptop ** akap), _pp(pe), _pp(peln), delp.p, pk.p, gz.p, hs.p,
                                              pt.p, _pp(pkz), C.c_int(int(CG))), "fv3_geopk")

    # -- vertical remap ---------------------------------------------------------------------------------
    def set_ak_bk(self, ak, bk):
        a = np.ascontiguousarray(ak, dtype=np.float64)
        b = np.ascontiguousarray(bk, dtype=np.float64)
        assert a.size == self.npz + 1 and b.size == self.npz + 1
        self.lib.check(self.lib.dll.fv3_set_ak_bk(self.h, a.ctypes.data_as(_dp), b.ctypes.data_as(_dp)), "fv3_set_ak_bk")

    def set_moist(self, par: dict | None, q_con=None, cappa=None):
        """moist_kappa / use_cond branches of the remap (fv_mapz.F90:212-219, :463-478, :806-811); None = off"""
        if par is None:
            self.lib.check(self.lib.dll.fv3_set_moist(self.h, None, None, None), "fv3_set_moist")
            return
        m = _MoistParams()
        for k, _ in _MoistParams._fields_:
            setattr(m, k, par.get(k, 0))
        self.lib.check(self.lib.dll.fv3_set_moist(self.h, C.byref(m), q_con.p if q_con is not None else None,
                                                  cappa.p if cappa is not None else None), "fv3_set_moist")

    def lagrangian_to_eulerian(self, par: dict, ps, pe, delp, pkz, pk, u, v, w, delz, pt, q, peln, omga, ws):
        """model/fv_mapz.F90:56 Lagrangian_to_Eulerian"""
        s = _RemapParams()
        for k in ["last_step", "hydrostatic", "adiabatic", "nq", "kord_mt", "kord_wz", "kord_tm", "sphum", "akap", "ptop",
                  "rdgas", "grav", "cv_air", "r_vir", "cp", "t_min"]:
            setattr(s, k, par[k])
        s.fill = int(par.get("fill", 0))
        kt = np.ascontiguousarray(par.get("kord_tr", []), dtype=np.int32)
        self.lib.check(self.lib.dll.fv3_lagrangian_to_eulerian(
            self.h, C.byref(s), kt.ctypes.data_as(_ip) if kt.size else None, ps.p, pe.p, delp.p, pkz.p, pk.p, u.p, v.p,
            _pp(w), _pp(delz), pt.p, _pp(q), peln.p, omga.p, _pp(ws)), "fv3_lagrangian_to_eulerian")

    @staticmethod
    def _remap_params(par: dict, last_step=None):
        s = _RemapParams()
        for k in ["last_step", "hydrostatic", "adiabatic", "nq", "kord_mt", "kord_wz", "kord_tm", "sphum", "akap", "ptop",
                  "rdgas", "grav", "cv_air", "r_vir", "cp", "t_min"]:
            setattr(s, k, par.get(k, 0) if k == "last_step" else par[k])
        if last_step is not None:
            s.last_step = int(last_step)
        s.fill = int(par.get("fill", 0))
        return s

    # -- consv_te: compute_total_energy (fv_thermodynamics.F90:90) and the energy fixer of the last remap (fv_mapz.F90:643-821)
    def compute_total_energy(self, par: dict, moist_phys, u, v, w, delz, pt, delp, q, qc, pe, peln, phis, te_2d):
        s = self._remap_params(par)
        self.lib.check(self.lib.dll.fv3_compute_total_energy(
            self.h, C.byref(s), C.c_int(int(moist_phys)), u.p, v.p, _pp(w), _pp(delz), pt.p, delp.p, _pp(q), _pp(qc), _pp(pe),
            _pp(peln), phis.p, te_2d.p), "fv3_compute_total_energy")

    def energy_fixer_sums(self, par: dict, only_sums, u, v, w, delz, pt, delp, q, pe, peln, phis, pkz, pk, te0_2d, te_2d, zsum1,
                          zsum0):
        s = self._remap_params(par)
        self.lib.check(self.lib.dll.fv3_energy_fixer_sums(
            self.h, C.byref(s), C.c_int(int(only_sums)), u.p, v.p, _pp(w), _pp(delz), pt.p, delp.p, _pp(q), _pp(pe), _pp(peln),
            phis.p, pkz.p, _pp(pk), _pp(te0_2d), _pp(te_2d), zsum1.p, _pp(zsum0)), "fv3_energy_fixer_sums")

    def ordered_sum(self, values) -> float:
        """g_sum(..., reproduce=.true.) of host values over the ranks of the context's communicator (fv3_ordered_sum)"""
        a = np.ascontiguousarray(values, dtype=np.float64).ravel()
        out = C.c_double(0.0)
        self.lib.check(self.lib.dll.fv3_ordered_sum(self.h, a.ctypes.data_as(_dp), C.c_size_t(a.size), C.byref(out)), "fv3_ordered_sum")
        return out.value

    def remap_finish(self, par: dict, dtmp, pt, pkz, q):
        s = self._remap_params(par)
        self.lib.check(self.lib.dll.fv3_remap_finish(self.h, C.byref(s), C.c_double(dtmp), pt.p, pkz.p, _pp(q)), "fv3_remap_finish")

    # -- tracer_2d ---------------------------------------------------------------------------------------
    def tracer_2d_prep(self, q_split, cx, cy, xfx, yfx) -> np.ndarray:
        cmax = np.zeros(self.npz)
        self.lib.check(self.lib.dll.fv3_tracer_2d_prep(self.h, C.c_int(q_split), cx.p, cy.p, xfx.p, yfx.p,
                                                       cmax.ctypes.data_as(_dp)), "fv3_tracer_2d_prep")
        return cmax

    def tracer_2d_scale(self, frac, cx, xfx, mfx, cy, yfx, mfy):
        f = np.ascontiguousarray(frac, dtype=np.float64)
        self.lib.check(self.lib.dll.fv3_tracer_2d_scale(self.h, f.ctypes.data_as(_dp), cx.p, xfx.p, mfx.p, cy.p, yfx.p,
                                                        mfy.p), "fv3_tracer_2d_scale")

    def tracer_2d_step(self, it, nsplt, ksplt, nq, hord, nord_tr, trdm, q, q_out, dp1, dp1_out, mfx, mfy, cx, cy, xfx, yfx):
        ks = np.ascontiguousarray(ksplt, dtype=np.int32)
        self.lib.check(self.lib.dll.fv3_tracer_2d_step(self.h, C.c_int(it), C.c_int(nsplt), ks.ctypes.data_as(_ip),
                                                       C.c_int(nq), C.c_int(hord), C.c_int(nord_tr), C.c_double(trdm),
                                                       q.p, q_out.p, dp1.p, dp1_out.p, mfx.p, mfy.p, cx.p, cy.p, xfx.p,
                                                       yfx.p), "fv3_tracer_2d_step")

    def set_remap_te(self, on, hs=None, te=None):
        """flagstruct%remap_te (fv_mapz.F90:232-286, :348-360, :576-619): hs = phis (A), te: A x npz work array"""
        self.lib.check(self.lib.dll.fv3_set_remap_te(self.h, C.c_int(int(bool(on))), hs.p if on else None, te.p if on else None),
                       "fv3_set_remap_te")

    def d_sw_inline_q(self, nq, hord_tr, nord_t, damp_t, q, q_out, delp_old, delp_new, fx, fy, crx, cry, xfx, yfx):
        """sw_core.F90:1020-1043: the tracers of one acoustic substep (after d_sw of the same substep, see the header)"""
        self.lib.check(self.lib.dll.fv3_d_sw_inline_q(self.h, C.c_int(nq), C.c_int(hord_tr), C.c_int(nord_t), C.c_double(damp_t), q.p,
                                                      q_out.p, delp_old.p, delp_new.p, fx.p, fy.p, crx.p, cry.p, xfx.p, yfx.p),
                       "fv3_d_sw_inline_q")

    def flux_accum(self, mfx, mfy, fx, fy):
        self.lib.check(self.lib.dll.fv3_flux_accum(self.h, mfx.p, mfy.p, fx.p, fy.p), "fv3_flux_accum")

    def fill2d_mass(self, nk, q, delp, qt, q_offset=0):
        """fv_fill.F90:228-235; q_offset: element offset of the tracer inside an A x npz x nq array"""
        qp = C.cast(_vp(q.ptr + 8 * q_offset), _dp)
        self.lib.check(self.lib.dll.fv3_fill2d_mass(self.h, C.c_int(nk), qp, delp.p, qt.p), "fv3_fill2d_mass")

    def fill2d_apply(self, nk, qt, delp, q, q_offset=0):
        """fv_fill.F90:238-256"""
        qp = C.cast(_vp(q.ptr + 8 * q_offset), _dp)
        self.lib.check(self.lib.dll.fv3_fill2d_apply(self.h, C.c_int(nk), qt.p, delp.p, qp), "fv3_fill2d_apply")

    def omga_update(self, rdt, ptop, pe, delp_before, omga):
        """dyn_core.F90:1182-1191: omga = (pe - pem)*rdt on the last substep (local part, see the header)"""
        self.lib.check(self.lib.dll.fv3_omga_update(self.h, C.c_double(rdt), C.c_double(ptop), pe.p, delp_before.p, omga.p),
                       "fv3_omga_update")

    def pt_to_theta_v(self, hydrostatic, zvir, kappa, rdgas, grav, pt, delp, delz, qv, pkz):
        """fv_dynamics.F90:284-329, :379-399: T -> theta_v before the k_split loop"""
        self.lib.check(self.lib.dll.fv3_pt_to_theta_v(
            self.h, C.c_int(int(hydrostatic)), C.c_double(zvir), C.c_double(kappa), C.c_double(rdgas), C.c_double(grav),
            pt.p, delp.p if delp is not None else None, delz.p if delz is not None else None,
            C.cast(_vp(getattr(qv, "ptr", qv)), _dp) if qv is not None else None, pkz.p), "fv3_pt_to_theta_v")

    def c2l(self, c2l_ord, u, v, ua, va):
        """cubed_to_latlon (fv_grid_utils.F90:2319), grid_type = 4 branches; ord 4 needs the halo of u, v"""
        self.lib.check(self.lib.dll.fv3_c2l(self.h, C.c_int(int(c2l_ord)), u.p, v.p, ua.p, va.p), "fv3_c2l")

    def rayleigh_u2f(self, kmax, hydrostatic, u, v, w, ua, va, u2f):
        """Rayleigh_Friction up to the halo update of u2f (fv_dynamics.F90:1186-1205)"""
        self.lib.check(self.lib.dll.fv3_rayleigh_u2f(self.h, C.c_int(int(kmax)), C.c_int(int(hydrostatic)), u.p, v.p,
                                                     w.p if w is not None else None, ua.p, va.p, u2f.p),
                       "fv3_rayleigh_u2f")

    def rayleigh_apply(self, kmax, conserve, hydrostatic, cp, rg, ptop, pm, rf, u2f, pt, delz, u, v, w):
        """Rayleigh_Friction after the halo update of u2f (fv_dynamics.F90:1211-1260); pm, rf: host arrays"""
        pm = np.ascontiguousarray(pm, dtype=np.float64)
        rf = np.ascontiguousarray(rf, dtype=np.float64)
        self.lib.check(self.lib.dll.fv3_rayleigh_apply(
            self.h, C.c_int(int(kmax)), C.c_int(int(conserve)), C.c_int(int(hydrostatic)), C.c_double(cp),
            C.c_double(rg), C.c_double(ptop), pm.ctypes.data_as(_dp), rf.ctypes.data_as(_dp), u2f.p, pt.p,
            delz.p if delz is not None else None, u.p, v.p, w.p if w is not None else None), "fv3_rayleigh_apply")

    def rayleigh_super(self, kmax, conserve, hydrostatic, cp, rg, ptop, pm, rf, ua, va, pt, u, v, w, u00=None, v00=None):
        """Rayleigh_Super after cubed_to_latlon (fv_dynamics.F90:1044-1121; grid_type < 4); pm, rf: host arrays"""
        pm = np.ascontiguousarray(pm, dtype=np.float64)
        rf = np.ascontiguousarray(rf, dtype=np.float64)
        self.lib.check(self.lib.dll.fv3_rayleigh_super(
            self.h, C.c_int(int(kmax)), C.c_int(int(conserve)), C.c_int(int(hydrostatic)), C.c_double(cp),
            C.c_double(rg), C.c_double(ptop), pm.ctypes.data_as(_dp), rf.ctypes.data_as(_dp), ua.p, va.p, pt.p, u.p, v.p,
            w.p if w is not None else None, u00.p if u00 is not None else None, v00.p if v00 is not None else None),
            "fv3_rayleigh_super")

    def compute_aam(self, radius, omega, agrav, ptop, coslat, ua, delp, aam, m_fac, ps):
        """compute_aam (fv_dynamics.F90:1266-1314) after c2l(2, ...): aam, m_fac (CC), ps (A)"""
        self.lib.check(self.lib.dll.fv3_compute_aam(self.h, C.c_double(radius), C.c_double(omega), C.c_double(agrav), C.c_double(ptop),
                                                    coslat.p, ua.p, delp.p, aam.p, m_fac.p, ps.p), "fv3_compute_aam")

    def consv_am_apply(self, u00, l2c_u, l2c_v, u, v):
        """fv_dynamics.F90:784-798: u += u00 l2c_u, v += u00 l2c_v"""
        self.lib.check(self.lib.dll.fv3_consv_am_apply(self.h, C.c_double(u00), l2c_u.p, l2c_v.p, u.p, v.p), "fv3_consv_am_apply")

    def adv_pe(self, ptop, ua, va, delp_before, omga):
        """adv_pe (dyn_core.F90:1195, :1529-1632): the advective term of omega on a cubed-sphere face"""
        self.lib.check(self.lib.dll.fv3_adv_pe(self.h, C.c_double(ptop), ua.p, va.p, delp_before.p, omga.p), "fv3_adv_pe")

    # ---- hydrostatic pressure gradient (dyn_core.F90:828-848, :1021, :1001-1010) ------------------------------------
    def divg2_ext(self, d_ext, delp, vt, divg2):
        self.lib.check(self.lib.dll.fv3_divg2_ext(self.h, C.c_double(d_ext), delp.p, vt.p, divg2.p), "fv3_divg2_ext")

    def split_p_grad(self, u, v, pp, gz, delp, pk, beta, dt, top_value, du, dv, gz_scale=1.0):
        """model/dyn_core.F90:1795 split_p_grad (beta: the caller's beta_d; du, dv: U / V x npz, zero before the first call)"""
        self.lib.check(self.lib.dll.fv3_split_p_grad(self.h, u.p, v.p, pp.p, gz.p, C.c_double(gz_scale), delp.p, pk.p, C.c_double(beta),
                                                     C.c_double(dt), C.c_double(top_value), du.p, dv.p), "fv3_split_p_grad")

    def grad1_p_update(self, divg2, u, v, pk, gz, dt, ptk, beta, du, dv):
        """model/dyn_core.F90:2033 grad1_p_update (divg2 None: zeros)"""
        self.lib.check(self.lib.dll.fv3_grad1_p_update(self.h, divg2.p if divg2 is not None else None, u.p, v.p, pk.p, gz.p,
                                                       C.c_double(dt), C.c_double(ptk), C.c_double(beta), du.p, dv.p),
                       "fv3_grad1_p_update")

    def one_grad_p(self, u, v, pk, gz, divg2, dt, ptk):
        self.lib.check(self.lib.dll.fv3_one_grad_p(self.h, u.p, v.p, pk.p, gz.p, divg2.p if divg2 is not None else None,
                                                   C.c_double(dt), C.c_double(ptk)), "fv3_one_grad_p")

    def one_grad_p_nh(self, u, v, pk, gz, divg2, delp, dt, ptop, gz_scale=1.0):
        """one_grad_p of the nonhydrostatic loop (beta < -0.1, dyn_core.F90:1029-1030): pk = the full pressure"""
        self.lib.check(self.lib.dll.fv3_one_grad_p_nh(self.h, u.p, v.p, pk.p, gz.p, divg2.p if divg2 is not None else None, delp.p,
                                                      C.c_double(dt), C.c_double(ptop), C.c_double(gz_scale)), "fv3_one_grad_p_nh")

    def copy_a_to_cc(self, src, dst):
        nk = int(np.prod(src.shape[2:])) if len(src.shape) > 2 else 1
        self.lib.check(self.lib.dll.fv3_copy_a_to_cc(self.h, src.p, dst.p, C.c_int(nk)), "fv3_copy_a_to_cc")

    # ---- dissipative heating after the substep loop (dyn_core.F90:798-803, :1300-1355) ---------------------------
    def heat_source_accum(self, heat_source, heat_s):
        self.lib.check(self.lib.dll.fv3_heat_source_accum(self.h, heat_source.p, heat_s.p), "fv3_heat_source_accum")

    def del2_cubed(self, q, cd: float, nmax: int):
        nk = int(np.prod(q.shape[2:])) if len(q.shape) > 2 else 1
        self.lib.check(self.lib.dll.fv3_del2_cubed(self.h, q.p, C.c_int(nk), C.c_double(cd), C.c_int(nmax)), "fv3_del2_cubed")

    def apply_heat_source(self, n_con, hydrostatic, bdt, delt_max, cp_air, cv_air, rdgas, grav, pt, heat_source, delp, delz,
                          pkz):
        self.lib.check(self.lib.dll.fv3_apply_heat_source(
            self.h, C.c_int(n_con), C.c_int(int(hydrostatic)), C.c_double(bdt), C.c_double(delt_max), C.c_double(cp_air),
            C.c_double(cv_air), C.c_double(rdgas), C.c_double(grav), pt.p, heat_source.p, delp.p,
            delz.p if delz is not None else None, pkz.p), "fv3_apply_heat_source")

    # ---- multi-rank halo exchange: pack / unpack (the transfers are halo.py's) ---------------------------
    def _halo_fields(self, fields):
        arr = (HaloField * len(fields))()
        for n, (dev, kind) in enumerate(fields):
            arr[n].field = C.cast(_vp(dev.ptr), _dp)
            arr[n].kind = {"A": 0, "U": 1, "V": 2, "B": 3}[kind]
            arr[n].nk = int(np.prod(dev.shape[2:])) if len(dev.shape) > 2 else 1
        return arr

    def halo_message_elems(self, fields):
        out = (C.c_size_t * 8)()
        self.lib.check(self.lib.dll.fv3_halo_message_elems(self.h, C.c_int(len(fields)), self._halo_fields(fields), out),
                       "fv3_halo_message_elems")
        return [int(v) for v in out]

    def halo_pack(self, fields, bufs):
        ptrs = (_dp * 8)(*[b.p for b in bufs])
        self.lib.check(self.lib.dll.fv3_halo_pack(self.h, C.c_int(len(fields)), self._halo_fields(fields), ptrs),
                       "fv3_halo_pack")

    def halo_periodic_group(self, fields):
        """one rank, doubly periodic: the halos of the group from its own opposite edges in one launch"""
        self.lib.check(self.lib.dll.fv3_halo_periodic_group(self.h, C.c_int(len(fields)), self._halo_fields(fields)),
                       "fv3_halo_periodic_group")

    def halo_unpack(self, fields, bufs):
        ptrs = (_dp * 8)(*[b.p for b in bufs])
        self.lib.check(self.lib.dll.fv3_halo_unpack(self.h, C.c_int(len(fields)), self._halo_fields(fields), ptrs),
                       "fv3_halo_unpack")

    # -- the exchange behind the C ABI (RCCL owned by the context): what a Fortran host without an RCCL binding uses ----------
    def comm_init(self, rank: int, nranks: int, unique_id: bytes | None = None) -> bytes:
        """fv3_comm_init; rank 0 of a one-rank run may leave unique_id out.  Returns the id used."""
        if unique_id is None:
            buf = (C.c_ubyte * 128)()
            self.lib.check(self.lib.dll.fv3_comm_get_unique_id(buf), "fv3_comm_get_unique_id")
            unique_id = bytes(buf)
        self.lib.check(self.lib.dll.fv3_comm_init(self.h, C.c_int(rank), C.c_int(nranks), (C.c_ubyte * 128)(*unique_id)),
                       "fv3_comm_init")
        return unique_id

    def halo_start(self, fields, to, frm):
        """start_group_halo_update: to[d] / frm[d] = the ranks at offsets d / -d, halo.DIRECTIONS order"""
        self.lib.check(self.lib.dll.fv3_halo_start(self.h, C.c_int(len(fields)), self._halo_fields(fields), (C.c_int * 8)(*to),
                                                   (C.c_int * 8)(*frm)), "fv3_halo_start")

    def halo_complete(self):
        self.lib.check(self.lib.dll.fv3_halo_complete(self.h), "fv3_halo_complete")

    def allreduce_max(self, a: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.float64).copy()
        self.lib.check(self.lib.dll.fv3_allreduce_max(self.h, a.ctypes.data_as(_dp), C.c_int(a.size)), "fv3_allreduce_max")
        return a

    def halo_fill_periodic(self, field: DeviceArray, kind: str):
        code = {"A": 0, "U": 1, "V": 2, "B": 3}[kind]
        nk = int(np.prod(field.shape[2:])) if len(field.shape) > 2 else 1
        self.lib.check(self.lib.dll.fv3_halo_fill_periodic(self.h, field.p, C.c_int(code), C.c_int(nk)),
                       "fv3_halo_fill_periodic")
