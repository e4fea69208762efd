"""Restart files of the dynamical core in the reference's layout, so that a state of this library can be compared with (``nccmp -d``,
the reference CI: .github/.parallelworks/run_test.sh:72-79) or started from a run of the reference.

Reference: tools/fv_io.F90 -- ``fv_io_register_restart`` (:206-442: which variables go into which file, on which axes),
``fv_io_register_axis`` (:122-196: the axis variables), ``fv_io_read_restart`` (:452-571) / ``fv_io_write_restart`` (:1176-1290: the
file names).  Per tile (``<N>`` = 1 .. 6 on the cubed sphere; ``.tile1`` also on a single-tile domain, :1217-1219):

  fv_core.res.nc                 xaxis_1 (npz + 1), Time          ak, bk                       (Time, xaxis_1)
  fv_core.res.tile<N>.nc         xaxis_1 (nx, CENTER)  xaxis_2 (nx + 1, EAST)  yaxis_1 (ny + 1, NORTH)  yaxis_2 (ny, CENTER)  zaxis_1 (npz)  Time
                                 u                             (Time, zaxis_1, yaxis_1, xaxis_1)     D-grid u on (is:ie, js:je+1)
                                 v                             (Time, zaxis_1, yaxis_2, xaxis_2)     D-grid v on (is:ie+1, js:je)
                                 W, DZ (nonhydrostatic), T, delp   (Time, zaxis_1, yaxis_2, xaxis_1)
                                 phis                          (Time, yaxis_2, xaxis_1)
                                 ua, va (agrid_vel_rst)        (Time, zaxis_1, yaxis_2, xaxis_1)
  fv_tracer.res.tile<N>.nc       xaxis_1 (nx)  yaxis_1 (ny)  zaxis_1 (npz)  Time     one variable per tracer name (Time, zaxis_1, yaxis_1, xaxis_1)
  fv_srf_wnd.res.tile<N>.nc      xaxis_1 (nx)  yaxis_1 (ny)  Time                    u_srf, v_srf (Time, yaxis_1, xaxis_1)

(netCDF lists the dimensions slowest first: the Fortran arrays are (x, y, z, Time).)  Every variable carries ``long_name`` = its name and
``units`` = "none"; the axis variables are doubles 1 .. n with ``cartesian_axis`` X / Y / Z / T, Time has units "time level" and the
value 1.  ``T`` is the temperature ``fv_dynamics`` leaves in ``pt`` after its last step, not the potential temperature of the loop.

FMS's fms2_io (release 2024.03, a dependency outside the reference tree) adds a ``checksum`` attribute to every restart variable: the
wrap-around 64-bit integer sum of the bit patterns of the values (``mpp_chksum``), as 16 upper-case hexadecimal digits; it is written
here by that published rule and compared on reading unless ``ignore_checksum`` (flagstruct%ignore_rst_cksum).  PARITY UNPINNED: no file
written by the reference is available to hold this writer to.

The files are netCDF classic with 64-bit offsets (``scipy.io.netcdf_file(version=2)``: the image has no netCDF4 library; fms2_io's
"64bit" format is the same container).  Host-side code: the arrays are numpy arrays in the library's field layout (layout.py), halos
included or not -- only the compute domain is written."""
from __future__ import annotations

import os

import numpy as np
from scipy.io import netcdf_file

from .layout import Bounds


def fms_checksum(a: np.ndarray) -> str:
    """mpp_chksum of a real(8) array as fms2_io writes it: sum of the values' bit patterns as 64-bit integers, modulo 2**64"""
    bits = np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)
    # Fortran's (Z16) edit descriptor: right-justified in 16 columns, BLANK padded (an all-zero field reads "               0")
    return "%16X" % (int(np.sum(bits, dtype=np.uint64)) & 0xFFFFFFFFFFFFFFFF)


def _checksum_value(text) -> int:
    """the number a checksum attribute holds, however it is padded (blanks of (Z16), zeros of older files of this writer)"""
    t = (text.decode() if isinstance(text, bytes) else str(text)).strip()
    return int(t or "0", 16)


def _axis(f, name: str, n, cart: str, units: str = "none"):
    f.createDimension(name, n)
    v = f.createVariable(name, "d", (name,))
    v.long_name = name
    v.units = units
    v.cartesian_axis = cart
    return v


def _field(f, name: str, dims, data: np.ndarray):
    v = f.createVariable(name, "d", ("Time",) + dims)
    v.long_name = name
    v.units = "none"
    v.checksum = fms_checksum(data)
    v[0] = data
    return v


def _compute(bd: Bounds, a: np.ndarray, kind: str, i1: int, j1: int) -> np.ndarray:
    """the compute domain (is:i1, js:j1) of an array that either carries the halo of its kind or is exactly that domain"""
    want = (i1 - bd.is_ + 1, j1 - bd.js + 1)
    if a.shape[:2] == want:
        return a
    return bd.view(a, kind, bd.is_, i1, bd.js, j1)


def _nc_order(a: np.ndarray) -> np.ndarray:
    """(x, y[, z]) Fortran arrays -> ([z, ]y, x) C arrays, the order of the netCDF dimensions"""
    return np.ascontiguousarray(np.transpose(a))


def tile_suffix(tile: int | None) -> str:
    return "" if tile is None else f".tile{tile}"


def write_core_levels(directory: str, ak, bk, prefix: str = ""):
    """fv_core.res.nc (fv_io.F90:249-282)"""
    ak, bk = np.asarray(ak, dtype=np.float64), np.asarray(bk, dtype=np.float64)
    os.makedirs(directory, exist_ok=True)
    with netcdf_file(os.path.join(directory, f"{prefix}fv_core.res.nc"), "w", version=2) as f:
        f.createDimension("Time", None)
        x = _axis(f, "xaxis_1", ak.size, "X")
        t = f.createVariable("Time", "d", ("Time",))
        t.long_name, t.units, t.cartesian_axis = "Time", "time level", "T"
        x[:] = np.arange(1, ak.size + 1, dtype=np.float64)
        _field(f, "ak", ("xaxis_1",), ak)
        _field(f, "bk", ("xaxis_1",), bk)
        t[0] = 1.0


def write_tile(directory: str, bd: Bounds, npz: int, state: dict, tile: int | None = 1, hydrostatic: bool = False, tracers: dict | None = None,
               srf_wnd: tuple | None = None, agrid_winds: bool = False, prefix: str = ""):
    """fv_core.res.tile<N>.nc, and fv_tracer.res.tile<N>.nc / fv_srf_wnd.res.tile<N>.nc when `tracers` ({name: array}) / `srf_wnd`
    ((u_srf, v_srf)) are given.  state: u, v, pt (= T), delp, phis, and w, delz unless hydrostatic (ua, va with agrid_winds); bd: the
    bounds of the tile (a whole tile: layout 1 x 1, or the tile assembled from its blocks by the caller)."""
    os.makedirs(directory, exist_ok=True)
    nx, ny = bd.nx, bd.ny
    sfx = tile_suffix(tile)
    with netcdf_file(os.path.join(directory, f"{prefix}fv_core.res{sfx}.nc"), "w", version=2) as f:
        f.createDimension("Time", None)
        axes = [_axis(f, "xaxis_1", nx, "X"), _axis(f, "xaxis_2", nx + 1, "X"), _axis(f, "yaxis_1", ny + 1, "Y"), _axis(f, "yaxis_2", ny, "Y"),
                _axis(f, "zaxis_1", npz, "Z")]
        t = f.createVariable("Time", "d", ("Time",))
        t.long_name, t.units, t.cartesian_axis = "Time", "time level", "T"
        for a in axes:
            a[:] = np.arange(1, a.shape[0] + 1, dtype=np.float64)
        c4 = ("zaxis_1", "yaxis_2", "xaxis_1")
        if agrid_winds:
            _field(f, "ua", c4, _nc_order(_compute(bd, state["ua"], "A", bd.ie, bd.je)))
            _field(f, "va", c4, _nc_order(_compute(bd, state["va"], "A", bd.ie, bd.je)))
        _field(f, "u", ("zaxis_1", "yaxis_1", "xaxis_1"), _nc_order(_compute(bd, state["u"], "U", bd.ie, bd.je + 1)))
        _field(f, "v", ("zaxis_1", "yaxis_2", "xaxis_2"), _nc_order(_compute(bd, state["v"], "V", bd.ie + 1, bd.je)))
        if not hydrostatic:
            _field(f, "W", c4, _nc_order(_compute(bd, state["w"], "A", bd.ie, bd.je)))
            _field(f, "DZ", c4, _nc_order(_compute(bd, state["delz"], "A", bd.ie, bd.je)))
        _field(f, "T", c4, _nc_order(_compute(bd, state["pt"], "A", bd.ie, bd.je)))
        _field(f, "delp", c4, _nc_order(_compute(bd, state["delp"], "A", bd.ie, bd.je)))
        _field(f, "phis", ("yaxis_2", "xaxis_1"), _nc_order(_compute(bd, state["phis"], "A", bd.ie, bd.je)))
        t[0] = 1.0
    for fname, fields, has_z in (("fv_tracer.res", tracers, True), ("fv_srf_wnd.res", dict(zip(("u_srf", "v_srf"), srf_wnd)) if srf_wnd else None, False)):
        if not fields:
            continue
        with netcdf_file(os.path.join(directory, f"{prefix}{fname}{sfx}.nc"), "w", version=2) as f:
            f.createDimension("Time", None)
            axes = [_axis(f, "xaxis_1", nx, "X"), _axis(f, "yaxis_1", ny, "Y")] + ([_axis(f, "zaxis_1", npz, "Z")] if has_z else [])
            t = f.createVariable("Time", "d", ("Time",))
            t.long_name, t.units, t.cartesian_axis = "Time", "time level", "T"
            for a in axes:
                a[:] = np.arange(1, a.shape[0] + 1, dtype=np.float64)
            for name, arr in fields.items():
                _field(f, name, (("zaxis_1",) if has_z else ()) + ("yaxis_1", "xaxis_1"), _nc_order(_compute(bd, arr, "A", bd.ie, bd.je)))
            t[0] = 1.0


def _read_var(f, name: str, ignore_checksum: bool) -> np.ndarray:
    v = f.variables[name]
    data = np.array(v[0], dtype=np.float64)           # a copy: the file is memory-mapped
    if not ignore_checksum and hasattr(v, "checksum"):
        want = v.checksum.decode() if isinstance(v.checksum, bytes) else str(v.checksum)
        got = fms_checksum(data)
        if _checksum_value(got) != _checksum_value(want):   # numerically: the padding of the attribute differs between writers
            raise ValueError(f"restart variable {name}: checksum {got} of the data is not the file's {want} (ignore_rst_cksum to read it anyway)")
    return np.asfortranarray(np.transpose(data))


def read_core_levels(directory: str, prefix: str = "", ignore_checksum: bool = False):
    with netcdf_file(os.path.join(directory, f"{prefix}fv_core.res.nc"), "r", mmap=False) as f:
        return _read_var(f, "ak", ignore_checksum), _read_var(f, "bk", ignore_checksum)


def read_tile(directory: str, bd: Bounds, npz: int, tile: int | None = 1, hydrostatic: bool = False, tracer_names=None, prefix: str = "",
              ignore_checksum: bool = False, with_halo: bool = True) -> dict:
    """-> {u, v, pt, delp, phis[, w, delz][, ua, va][, q: {name: array}][, u_srf, v_srf]}: arrays in the library's layout (halo'd, halo
    zero: the model's first halo update fills it) or, with_halo=False, the compute-domain arrays as stored.  Missing optional files
    (tracers, surface winds) are skipped like the reference does (fv_io.F90:531-551); a missing variable of fv_core.res raises."""
    sfx = tile_suffix(tile)
    out = {}

    def place(data, kind, i1, j1):
        if not with_halo:
            return data
        full = bd.zeros(kind, data.shape[2]) if data.ndim == 3 else bd.zeros(kind)
        bd.view(full, kind, bd.is_, i1, bd.js, j1)[...] = data
        return full
    with netcdf_file(os.path.join(directory, f"{prefix}fv_core.res{sfx}.nc"), "r", mmap=False) as f:
        if f.dimensions["xaxis_1"] != bd.nx or f.dimensions["yaxis_2"] != bd.ny or f.dimensions["zaxis_1"] != npz:
            raise ValueError(f"fv_core.res{sfx}.nc is {f.dimensions['xaxis_1']} x {f.dimensions['yaxis_2']} x {f.dimensions['zaxis_1']}, "
                             f"the model {bd.nx} x {bd.ny} x {npz}")
        out["u"] = place(_read_var(f, "u", ignore_checksum), "U", bd.ie, bd.je + 1)
        out["v"] = place(_read_var(f, "v", ignore_checksum), "V", bd.ie + 1, bd.je)
        for key, name in (("pt", "T"), ("delp", "delp"), ("phis", "phis")) + ((() if hydrostatic else (("w", "W"), ("delz", "DZ")))):
            d = _read_var(f, name, ignore_checksum)
            out[key] = d if key == "delz" and with_halo else place(d, "A", bd.ie, bd.je)     # delz has no halo in the model (fv_arrays.F90)
        for name in ("ua", "va"):
            if name in f.variables:
                out[name] = place(_read_var(f, name, ignore_checksum), "A", bd.ie, bd.je)
    p = os.path.join(directory, f"{prefix}fv_tracer.res{sfx}.nc")
    if os.path.exists(p):
        with netcdf_file(p, "r", mmap=False) as f:
            names = tracer_names if tracer_names is not None else [n for n in f.variables if n not in ("xaxis_1", "yaxis_1", "zaxis_1", "Time")]
            out["q"] = {n: place(_read_var(f, n, ignore_checksum), "A", bd.ie, bd.je) for n in names if n in f.variables}
    p = os.path.join(directory, f"{prefix}fv_srf_wnd.res{sfx}.nc")
    if os.path.exists(p):
        with netcdf_file(p, "r", mmap=False) as f:
            for n in ("u_srf", "v_srf"):
                out[n] = place(_read_var(f, n, ignore_checksum), "A", bd.ie, bd.je)
    return out


def assemble_tile(blocks, npx: int, npy: int, npz: int, names=("u", "v", "w", "delz", "pt", "delp", "phis")):
    """[(Bounds, state dict)] of the ranks that share a tile -> (Bounds of the tile, state of the tile): what FMS's io domain does when
    several ranks write one file.  The shared staggered rows / columns of u, v are taken from the block that owns them."""
    bd_t = Bounds(1, npx - 1, 1, npy - 1)
    kinds = {"u": "U", "v": "V"}
    out = {}
    for n in names:
        if not all(n in st for _, st in blocks):
            continue
        kind = kinds.get(n, "A")
        first = blocks[0][1][n]
        full = bd_t.zeros(kind, npz) if first.ndim == 3 else bd_t.zeros(kind)
        for bd, st in blocks:
            i1 = bd.ie + (1 if kind == "V" else 0)
            j1 = bd.je + (1 if kind == "U" else 0)
            bd_t.view(full, kind, bd.is_, i1, bd.js, j1)[...] = _compute(bd, st[n], kind, i1, j1)
        out[n] = full
    return bd_t, out
