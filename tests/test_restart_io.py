"""restart_io.py: the reference's restart files (tools/fv_io.F90:206-571) -- variable set, axes and dimension order, a bit-for-bit
round trip, the tile assembled from the blocks of several ranks, the FMS checksum attribute, and a restarted run that continues
bit for bit (what the reference CI checks with nccmp, .github/.parallelworks/run_test.sh:72-79)."""
import os
import subprocess

import numpy as np
import pytest
from scipy.io import netcdf_file

from gfdl_atmos_cubed_sphere_amd import restart_io as RIO
from gfdl_atmos_cubed_sphere_amd.layout import Bounds

HERE = os.path.dirname(os.path.abspath(__file__))


def _state(bd, npz, seed=5, nq=3):
    rng = np.random.default_rng(seed)
    st = {"u": np.asfortranarray(rng.normal(0, 10, bd.shape("U", npz))), "v": np.asfortranarray(rng.normal(0, 10, bd.shape("V", npz))),
          "w": np.asfortranarray(rng.normal(0, 1, bd.shape("A", npz))), "pt": np.asfortranarray(rng.uniform(200, 300, bd.shape("A", npz))),
          "delp": np.asfortranarray(rng.uniform(100, 2000, bd.shape("A", npz))), "phis": np.asfortranarray(rng.uniform(0, 1e4, bd.shape("A"))),
          "delz": np.asfortranarray(-rng.uniform(50, 900, bd.shape("CC", npz)))}
    q = {n: np.asfortranarray(rng.uniform(0, 1e-2, bd.shape("A", npz))) for n in ("sphum", "liq_wat", "o3mr")[:nq]}
    srf = (np.asfortranarray(rng.normal(0, 5, bd.shape("A"))), np.asfortranarray(rng.normal(0, 5, bd.shape("A"))))
    return st, q, srf


def test_files_axes_and_variables_are_the_reference_s(tmp_path):
    """fv_io_register_restart (:206-442): file names, axis names and sizes, which variable sits on which axes (netCDF order = the
    Fortran order reversed), the axis variables and their attributes"""
    bd, npz = Bounds(1, 12, 1, 8), 5
    st, q, srf = _state(bd, npz)
    ak, bk = np.linspace(300.0, 0.0, npz + 1), np.linspace(0.0, 1.0, npz + 1)
    d = str(tmp_path)
    RIO.write_core_levels(d, ak, bk)
    RIO.write_tile(d, bd, npz, st, tile=3, tracers=q, srf_wnd=srf)
    assert sorted(os.listdir(d)) == ["fv_core.res.nc", "fv_core.res.tile3.nc", "fv_srf_wnd.res.tile3.nc", "fv_tracer.res.tile3.nc"]
    with netcdf_file(os.path.join(d, "fv_core.res.nc"), mmap=False) as f:
        assert f.dimensions == {"Time": None, "xaxis_1": npz + 1}
        assert f.variables["ak"].dimensions == ("Time", "xaxis_1") and f.variables["bk"].dimensions == ("Time", "xaxis_1")
        assert np.array_equal(f.variables["ak"][0], ak) and np.array_equal(f.variables["xaxis_1"][:], np.arange(1, npz + 2))
        assert f.variables["Time"].units == b"time level" and f.variables["Time"][0] == 1.0
    with netcdf_file(os.path.join(d, "fv_core.res.tile3.nc"), mmap=False) as f:
        assert f.dimensions == {"Time": None, "xaxis_1": 12, "xaxis_2": 13, "yaxis_1": 9, "yaxis_2": 8, "zaxis_1": npz}
        dims = {n: v.dimensions for n, v in f.variables.items()}
        assert dims["u"] == ("Time", "zaxis_1", "yaxis_1", "xaxis_1")          # dim_names_4d
        assert dims["v"] == ("Time", "zaxis_1", "yaxis_2", "xaxis_2")          # dim_names_4d2
        for n in ("W", "DZ", "T", "delp"):
            assert dims[n] == ("Time", "zaxis_1", "yaxis_2", "xaxis_1")      # dim_names_4d3
        assert dims["phis"] == ("Time", "yaxis_2", "xaxis_1")                  # dim_names_3d
        assert set(dims) == {"xaxis_1", "xaxis_2", "yaxis_1", "yaxis_2", "zaxis_1", "Time", "u", "v", "W", "DZ", "T", "delp", "phis"}
        assert f.variables["xaxis_2"].cartesian_axis == b"X" and f.variables["zaxis_1"].cartesian_axis == b"Z"
        assert f.variables["T"].long_name == b"T" and f.variables["T"].units == b"none"
        # u(x, y, z) of the model is [z, y, x] in the file
        assert f.variables["u"][0][2, 5, 7] == bd.view(st["u"], "U", 1, 12, 1, 9)[7, 5, 2]
    with netcdf_file(os.path.join(d, "fv_tracer.res.tile3.nc"), mmap=False) as f:
        assert f.dimensions == {"Time": None, "xaxis_1": 12, "yaxis_1": 8, "zaxis_1": npz}
        assert f.variables["liq_wat"].dimensions == ("Time", "zaxis_1", "yaxis_1", "xaxis_1")
    with netcdf_file(os.path.join(d, "fv_srf_wnd.res.tile3.nc"), mmap=False) as f:
        assert f.variables["u_srf"].dimensions == ("Time", "yaxis_1", "xaxis_1")


@pytest.mark.parametrize("hydrostatic", [False, True])
def test_round_trip_is_bit_for_bit(tmp_path, hydrostatic):
    bd, npz = Bounds(1, 10, 1, 7), 4
    st, q, srf = _state(bd, npz, seed=11)
    d = str(tmp_path)
    RIO.write_core_levels(d, np.arange(npz + 1.0), np.arange(npz + 1.0) / npz, prefix="20260101.")
    RIO.write_tile(d, bd, npz, st, tile=1, hydrostatic=hydrostatic, tracers=q, srf_wnd=srf, prefix="20260101.")
    back = RIO.read_tile(d, bd, npz, tile=1, hydrostatic=hydrostatic, prefix="20260101.")
    ak, bk = RIO.read_core_levels(d, prefix="20260101.")
    assert np.array_equal(ak, np.arange(npz + 1.0)) and np.array_equal(bk, np.arange(npz + 1.0) / npz)
    for n, kind, i1, j1 in (("u", "U", bd.ie, bd.je + 1), ("v", "V", bd.ie + 1, bd.je), ("pt", "A", bd.ie, bd.je), ("delp", "A", bd.ie, bd.je),
                            ("phis", "A", bd.ie, bd.je)) + (() if hydrostatic else (("w", "A", bd.ie, bd.je),)):
        assert np.array_equal(bd.view(back[n], kind, 1, i1, 1, j1), bd.view(st[n], kind, 1, i1, 1, j1)), n
        halo = back[n].copy()
        bd.view(halo, kind, 1, i1, 1, j1)[...] = 0.0
        assert not halo.any()                                   # the halo is left for the model's first update
    assert ("w" in back) == (not hydrostatic)
    if not hydrostatic:
        assert np.array_equal(back["delz"], st["delz"])
    for n in q:
        assert np.array_equal(bd.view(back["q"][n], "A", 1, bd.ie, 1, bd.je), bd.view(q[n], "A", 1, bd.ie, 1, bd.je))
    assert np.array_equal(bd.view(back["u_srf"], "A", 1, bd.ie, 1, bd.je), bd.view(srf[0], "A", 1, bd.ie, 1, bd.je))


def test_checksum_attribute(tmp_path):
    """fms2_io's checksum of a restart variable: the wrap-around 64-bit sum of the bit patterns, 16 hex digits; a file whose data no
    longer match it is refused unless ignore_rst_cksum"""
    a = np.array([1.0, -2.5, 3.0e300, -0.0])
    want = sum(int(x) for x in a.view(np.uint64)) % (1 << 64)
    assert int(RIO.fms_checksum(a), 16) == want and len(RIO.fms_checksum(a)) == 16
    # Fortran (Z16): blank padded -- an all-zero field (phis over the ocean) is 15 blanks and a 0, and reads back; so does a file whose
    # writer padded with zeros
    z = np.zeros((4, 3))
    assert RIO.fms_checksum(z) == " " * 15 + "0"
    assert RIO._checksum_value(b"000000000000000A") == RIO._checksum_value("               A") == 10
    bd, npz = Bounds(1, 6, 1, 5), 3
    st, _, _ = _state(bd, npz)
    d = str(tmp_path)
    RIO.write_tile(d, bd, npz, st)
    p = os.path.join(d, "fv_core.res.tile1.nc")
    with netcdf_file(p, "a", mmap=False) as f:
        f.variables["T"][0, 1, 2, 3] += 1.0
    with pytest.raises(ValueError, match="checksum"):
        RIO.read_tile(d, bd, npz)
    back = RIO.read_tile(d, bd, npz, ignore_checksum=True)
    assert back["pt"][3 + 3, 2 + 3, 1] == st["pt"][3 + 3, 2 + 3, 1] + 1.0


def test_tile_from_the_blocks_of_four_ranks(tmp_path):
    """a tile decomposed 2 x 2: the file is the whole tile (FMS's io domain), staggered rows / columns from their owners"""
    npx, npy, npz = 13, 9, 3
    bd_t = Bounds(1, npx - 1, 1, npy - 1)
    st, _, _ = _state(bd_t, npz, seed=3)
    blocks = []
    for iy in range(2):
        for ix in range(2):
            bd = Bounds(1 + ix * 6, 6 + ix * 6, 1 + iy * 4, 4 + iy * 4)
            loc = {}
            for n, kind in (("u", "U"), ("v", "V"), ("w", "A"), ("pt", "A"), ("delp", "A"), ("phis", "A")):
                ilo, ihi, jlo, jhi = bd.limits(kind)
                src = np.pad(st[n], [(0, 0)] * st[n].ndim)      # the tile's own halo'd array covers a block's halo only inside the tile
                full = bd.zeros(kind, npz) if st[n].ndim == 3 else bd.zeros(kind)
                i1, j1 = bd.ie + (kind == "V"), bd.je + (kind == "U")
                bd.view(full, kind, bd.is_, i1, bd.js, j1)[...] = bd_t.view(src, kind, bd.is_, i1, bd.js, j1)
                loc[n] = full
            loc["delz"] = np.asfortranarray(st["delz"][ix * 6:(ix + 1) * 6, iy * 4:(iy + 1) * 4, :])
            blocks.append((bd, loc))
    bd2, whole = RIO.assemble_tile(blocks, npx, npy, npz)
    d = str(tmp_path)
    RIO.write_tile(d, bd2, npz, whole)
    RIO.write_tile(os.path.join(d, "one"), bd_t, npz, st)
    a, b = open(os.path.join(d, "fv_core.res.tile1.nc"), "rb").read(), open(os.path.join(d, "one", "fv_core.res.tile1.nc"), "rb").read()
    assert a == b


def check_restarted_run_continues(lib, workdir, nx=16, ny=12, npz=8, nq=2):
    """two fv_dynamics calls in one go against one call, restart files, a fresh context started from them, the second call: the
    same final state bit for bit (the reference CI's restart-reproducibility check); nonhydrostatic, tracers"""
    import parity_common as P
    import parity_dyn as D
    import parity_nh as N
    from gfdl_atmos_cubed_sphere_amd.dyn_core import DynFlags
    from gfdl_atmos_cubed_sphere_amd.fv_dynamics import FvDynamics
    from gfdl_atmos_cubed_sphere_amd.layout import periodic_fill
    from gfdl_atmos_cubed_sphere_amd.lib import GRAV, KAPPA, RDGAS, Context
    emu = lib
    bd = Bounds(1, nx, 1, ny)
    g = P.make_grid(bd, False)
    st, _ = D.make_state(bd, npz)
    r = (bd.is_, bd.ie, bd.js, bd.je)
    ng = bd.ng
    th = bd.view(st["pt"], "A", *r)
    pkz0 = np.exp(KAPPA / (1.0 - KAPPA) * np.log((-RDGAS / GRAV) * bd.view(st["delp"], "A", *r) * th / st["delz"]))
    T = st["pt"].copy(order="F")
    T[ng:ng + nx, ng:ng + ny, :] = th * pkz0
    sig = np.linspace(0.0, 1.0, npz + 1) ** 1.5
    ak, bk = N.PTOP * (1.0 - sig), sig.copy()
    fl = DynFlags(n_split=2, ptop=N.PTOP)
    q0 = np.asfortranarray(np.random.default_rng(2).uniform(0, 1e-2, bd.shape("A", npz) + (nq,)))
    names = ["sphum", "o3mr"]

    def fresh():
        ctx = Context(g, npz, lib=emu)
        return ctx, FvDynamics(ctx, fl, ak, bk, nq=nq, k_split=2)

    def final(fv):
        out = fv.dc.get_state()
        out["q"] = fv.dc.d["q"].download()
        return out
    ctx, fv = fresh()
    fv.dc.set_state(st["u"], st["v"], st["w"], st["delp"], T, st["delz"], st["phis"])
    fv.set_tracers(q0)
    fv.step_from_temperature(8.0)
    mid = final(fv)
    d = str(workdir)
    RIO.write_core_levels(d, ak, bk)
    RIO.write_tile(d, bd, npz, dict(mid, phis=st["phis"]), tracers={n: mid["q"][:, :, :, i] for i, n in enumerate(names)})
    fv.step_from_temperature(8.0)
    straight = final(fv)
    ctx.close()
    # ---- the restarted run ----
    ak2, bk2 = RIO.read_core_levels(d)
    assert np.array_equal(ak2, ak) and np.array_equal(bk2, bk)
    rs = RIO.read_tile(d, bd, npz, tracer_names=names)
    for n, kind in (("u", "U"), ("v", "V"), ("w", "A"), ("delp", "A"), ("pt", "A"), ("phis", "A")):      # the model's first halo update
        a = rs[n]
        for k in range(a.shape[2] if a.ndim == 3 else 1):
            periodic_fill(bd, a[:, :, k] if a.ndim == 3 else a, kind)
    q1 = np.asfortranarray(np.stack([rs["q"][n] for n in names], axis=-1))
    for i in range(nq):
        for k in range(npz):
            periodic_fill(bd, q1[:, :, k, i], "A")
    ctx, fv = fresh()
    fv.dc.set_state(rs["u"], rs["v"], rs["w"], rs["delp"], rs["pt"], rs["delz"], rs["phis"])
    fv.set_tracers(q1)
    fv.step_from_temperature(8.0)
    again = final(fv)
    ctx.close()
    for n, kind, rr in (("u", "U", (1, nx, 1, ny + 1)), ("v", "V", (1, nx + 1, 1, ny)), ("w", "A", (1, nx, 1, ny)), ("delp", "A", (1, nx, 1, ny)),
                        ("pt", "A", (1, nx, 1, ny))):
        assert np.array_equal(bd.view(again[n], kind, *rr), bd.view(straight[n], kind, *rr)), n
    assert np.array_equal(again["delz"], straight["delz"])
    assert np.array_equal(bd.view(again["q"], "A", 1, nx, 1, ny), bd.view(straight["q"], "A", 1, nx, 1, ny))


def test_a_restarted_run_continues_bit_for_bit(tmp_path):
    """host-emulation build (the CPU suite)"""
    from gfdl_atmos_cubed_sphere_amd.lib import Fv3Lib
    subprocess.check_call(["make", "-C", os.path.join(HERE, "hostemu"), "-s"])
    check_restarted_run_continues(Fv3Lib(os.path.join(HERE, "hostemu", "libfv3_hostemu.so")), tmp_path)


@pytest.mark.gpu
def test_a_restarted_run_continues_bit_for_bit_on_the_gpu(tmp_path):
    """the same through the HIP library on the MI355X, at a size with several strips and segments of the marching kernels: the device
    state written as the reference's restart files (fv_core.res / fv_tracer.res per tile, tools/fv_io.F90:206-571), a fresh context
    started from the files, the run continued -- bit for bit the uninterrupted run"""
    from gfdl_atmos_cubed_sphere_amd import lib as L
    check_restarted_run_continues(L.load(), tmp_path, nx=130, ny=70, npz=12, nq=2)
