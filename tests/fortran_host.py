"""Build and run the Fortran host (gfdl_atmos_cubed_sphere_amd/fortran: fv3_mi355x_mod + fv3_host_mod + fv3_solo) and
compare what it leaves in the state with the Python host driving the same library through the same C ABI."""
import os
import shutil
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FDIR = os.path.join(ROOT, "gfdl_atmos_cubed_sphere_amd", "fortran")
CSRC = os.path.join(ROOT, "gfdl_atmos_cubed_sphere_amd", "csrc")


def fortran_compiler():
    fc = shutil.which("amdflang") or "/opt/rocm/bin/amdflang"
    return fc if os.path.exists(fc) else None


def _compile(workdir, name, files, libdir, libname):
    """amdflang once per (work directory, library): a test that runs several cases of one driver compiles it once"""
    fc = fortran_compiler()
    exe = os.path.join(str(workdir), name)
    stamp = exe + ".built_against"
    want = os.path.join(libdir, "lib" + libname + ".so")
    if os.path.exists(exe) and os.path.exists(stamp) and open(stamp).read() == want:
        return exe
    srcs = [os.path.join(FDIR, f) for f in files]
    subprocess.check_call([fc, "-O1", "-module-dir", str(workdir)] + srcs + ["-L" + libdir, "-l" + libname, "-Wl,-rpath," + libdir, "-o", exe])
    with open(stamp, "w") as f:
        f.write(want)
    return exe


def build_solo(workdir, libdir=CSRC, libname="fv3_mi355x"):
    """amdflang: interface module + host module + driver, linked against the product library"""
    return _compile(workdir, "fv3_solo", ('fv3_mi355x_mod.F90', 'fv3_host_mod.F90', 'fv3_solo.F90'), libdir, libname)


def write_input(path, bd, npz, nq, n_split, k_split, nsteps, last_step, dx, dy, f0, bdt, ptop, ak, bk, st, q, hydrostatic=False,
                d_con=0.0, d_ext=0.02, beta=0.0, inline_q=False, remap_te=False, moist=None, consv_am=None):
    with open(path, "wb") as f:
        np.array([bd.nx, bd.ny, npz, nq, n_split, k_split, nsteps, int(last_step), int(hydrostatic) + 2 * int(inline_q) + 4 * int(remap_te) +
                  (8 * int(moist["use_cond"]) + 16 * int(moist["moist_kappa"]) if moist else 0) + (32 if consv_am else 0)], dtype=np.int32).tofile(f)
        np.array([dx, dy, f0, bdt, ptop, d_con, d_ext, beta], dtype=np.float64).tofile(f)
        np.asarray(ak, dtype=np.float64).tofile(f)
        np.asarray(bk, dtype=np.float64).tofile(f)
        for n in ("u", "v", "w", "delp", "pt", "delz", "phis"):
            np.asfortranarray(st[n], dtype=np.float64).ravel(order="F").tofile(f)
        if nq:
            np.asfortranarray(q, dtype=np.float64).ravel(order="F").tofile(f)
        if moist:
            for n in ("q_con", "cappa"):
                np.asfortranarray(moist[n], dtype=np.float64).ravel(order="F").tofile(f)
        if consv_am:   # agrid(:,:,2) on (isd:ied, jsd:jed), gridstruct%l2c_u (is:ie, js:je+1), %l2c_v (is:ie+1, js:je), idiag%zxg (compute domain)
            ng = bd.ng
            np.asfortranarray(consv_am["lat"], dtype=np.float64).ravel(order="F").tofile(f)
            np.asfortranarray(consv_am["l2c_u"][ng:ng + bd.nx, ng:ng + bd.ny + 1], dtype=np.float64).ravel(order="F").tofile(f)
            np.asfortranarray(consv_am["l2c_v"][ng:ng + bd.nx + 1, ng:ng + bd.ny], dtype=np.float64).ravel(order="F").tofile(f)
            np.asfortranarray(consv_am["zxg"], dtype=np.float64).ravel(order="F").tofile(f)


def read_output(path, bd, npz, nq):
    out = {}
    with open(path, "rb") as f:
        for n, kind in (("u", "U"), ("v", "V"), ("w", "A"), ("delp", "A"), ("pt", "A"), ("delz", "CC")):
            shp = bd.shape(kind, npz)
            out[n] = np.fromfile(f, dtype=np.float64, count=int(np.prod(shp))).reshape(shp, order="F")
        if nq:
            shp = bd.shape("A", npz) + (nq,)
            out["q"] = np.fromfile(f, dtype=np.float64, count=int(np.prod(shp))).reshape(shp, order="F")
    return out


def check_fortran_host(lib, workdir, nx=40, ny=24, npz=10, nq=2, n_split=2, k_split=2, nsteps=2, bdt=8.0, host_comm=False,
                       hydrostatic=False, d_con=0.0, beta=0.0, inline_q=False, remap_te=False, use_cond=False, moist_kappa=False):
    """the same initial state through (a) the Python host and (b) the Fortran host: bit-identical states"""
    import parity_common as P
    import parity_dyn as D
    import parity_nh as N
    from gfdl_atmos_cubed_sphere_amd.dyn_core import DynFlags
    from gfdl_atmos_cubed_sphere_amd.fv_dynamics import FvDynamics
    from gfdl_atmos_cubed_sphere_amd.layout import Bounds
    from gfdl_atmos_cubed_sphere_amd.lib import Context
    bd = Bounds(1, nx, 1, ny)
    g = P.make_grid(bd, False)
    st, _ = D.make_state(bd, npz)
    sig = np.linspace(0.0, 1.0, npz + 1) ** 1.5
    ak, bk = N.PTOP * (1.0 - sig), sig.copy()
    rng = np.random.default_rng(5)
    q = np.asfortranarray(rng.uniform(0.0, 1.0, bd.shape("A", npz) + (nq,))) if nq else None
    fl = DynFlags(n_split=n_split, ptop=N.PTOP, hydrostatic=hydrostatic, d_con=d_con, beta=beta, inline_q=inline_q, use_cond=use_cond,
                  moist_kappa=moist_kappa)
    moist = None
    if use_cond or moist_kappa:        # six water species in tracers 1 .. 6 (small mixing ratios), some q_con and cappa to start from
        import parity_remap as R
        assert nq >= 6
        q[..., 0] *= 0.02
        q[..., 1:6] *= 0.002
        from gfdl_atmos_cubed_sphere_amd.lib import CP_AIR, RDGAS
        cvm, qc = N.np_moist_cv(q, dict(R.MOIST6, sphum=1), CP_AIR - RDGAS)     # q_con and cappa as fv_dynamics.F90:305-317 forms them
        moist = dict(use_cond=use_cond, moist_kappa=moist_kappa, q_con=np.asfortranarray(qc),
                     cappa=np.asfortranarray(RDGAS / (RDGAS + cvm / (1.0 + 0.6077 * q[..., 0]))))
    # ---- (a) Python host ----
    ctx = Context(g, npz, lib=lib)
    try:
        fv = FvDynamics(ctx, fl, ak, bk, nq=nq, k_split=k_split, remap_te=remap_te, moist=dict(R.MOIST6) if moist else None)
        fv.dc.set_state(st["u"], st["v"], st["w"], st["delp"], st["pt"], st["delz"], st["phis"])
        if moist:
            if use_cond:
                fv.dc.d["q_con"].upload(moist["q_con"])
            if moist_kappa:
                fv.dc.d["cappa"].upload(moist["cappa"])
        if nq:
            fv.set_tracers(q)
        for _ in range(nsteps):
            fv.step(bdt)
        d = fv.dc.d
        ref = {n: d[n].download() for n in (("u", "v", "delp", "pt") if hydrostatic else ("u", "v", "w", "delp", "pt", "delz"))}
        if nq:
            ref["q"] = d["q"].download()
    finally:
        ctx.close()
    # ---- (b) Fortran host ----
    exe = build_solo(workdir, libdir=os.path.dirname(lib.path), libname=os.path.basename(lib.path)[3:-3])
    fin, fout = os.path.join(str(workdir), "in.bin"), os.path.join(str(workdir), "out.bin")
    write_input(fin, bd, npz, nq, n_split, k_split, nsteps, False, 1000.0, 1000.0, float(g.m["f0"][0, 0]), bdt, N.PTOP, ak,
                bk, st, q, hydrostatic=hydrostatic, d_con=d_con, d_ext=fl.d_ext, beta=beta, inline_q=inline_q, remap_te=remap_te, moist=moist)
    env = dict(os.environ, **({"FV3_HOST_COMM": "1"} if host_comm else {}))
    r = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "geometry mode 2" in r.stdout, r.stdout
    assert ("fv3_halo_start" in r.stdout) == bool(host_comm), r.stdout
    got = read_output(fout, bd, npz, nq)
    i0, i1, j0, j1 = bd.is_, bd.ie, bd.js, bd.je
    rng_ = {"u": ("U", i0, i1, j0, j1 + 1), "v": ("V", i0, i1 + 1, j0, j1), "w": ("A", i0, i1, j0, j1),
            "delp": ("A", i0, i1, j0, j1), "pt": ("A", i0, i1, j0, j1)}
    for n in ref:
        if n in rng_:
            kind, *r4 = rng_[n]
            a, b = bd.view(got[n], kind, *r4), bd.view(ref[n], kind, *r4)
        elif n == "q":
            a, b = got[n][bd.ng:bd.ng + nx, bd.ng:bd.ng + ny], ref[n][bd.ng:bd.ng + nx, bd.ng:bd.ng + ny]
        else:
            a, b = got[n], ref[n]
        assert np.all(np.isfinite(a)), n
        assert np.array_equal(a, b), f"{n}: Fortran host and Python host differ (max abs {np.max(np.abs(a - b)):.3e})"
    return r.stdout


def build_refsig(workdir, libdir=CSRC, libname="fv3_mi355x"):
    """the reference-signature dyn_core (fv3_dyn_core_mod) + its driver"""
    return _compile(workdir, "fv3_solo_refsig", ('fv3_mi355x_mod.F90', 'fv3_host_mod.F90', 'fv3_sphere_mod.F90', 'fv3_dyn_core_mod.F90', 'fv3_solo_refsig.F90'), libdir, libname)


def check_fortran_refsig(lib, workdir, nx=24, ny=16, npz=8, n_split=3, nsteps=2, bdt=6.0, hydrostatic=False, d_con=0.0, beta=0.0, moist=False,
                         layout=(1, 1), fast_tau_w_sec=0.0, rf_fast_tau=0.0, registry=False, do_diss_est=False, fill_dp=False):
    """dyn_core called with the reference's argument list on host arrays (fv3_dyn_core_mod, driver fv3_solo_refsig) against the
    Python host's DynCore.run on the same state: u, v, w, delp, pt, delz, the accumulated mass fluxes / Courant numbers (and pkz
    when the heating or the hydrostatic branch writes it) bit-identical"""
    import parity_common as P
    import parity_dyn as D
    import parity_nh as N
    from gfdl_atmos_cubed_sphere_amd.dyn_core import DynCore, DynFlags
    from gfdl_atmos_cubed_sphere_amd.layout import Bounds
    from gfdl_atmos_cubed_sphere_amd.lib import Context
    bd = Bounds(1, nx, 1, ny)
    g = P.make_grid(bd, False)
    if do_diss_est:     # flagstruct%do_diss_est: the SKEB dissipation estimate summed over the substeps of every call (dyn_core.F90:805-811)
        import dataclasses
        g = dataclasses.replace(g, do_diss_est=True, prevent_diss_cooling=False)
    st, _ = D.make_state(bd, npz)
    sig = np.linspace(0.0, 1.0, npz + 1) ** 1.5
    ak, bk = N.PTOP * (1.0 - sig), sig.copy()
    if fill_dp:     # flagstruct%fill_dp: reference thicknesses that make about half of the cells of every layer `thin` (mix_dp acts)
        ak, bk = D.thin_akbk(bd, npz, st["delp"])
    # fast_tau_w_sec / RF_fast (dyn_core.F90:536, :940, :1057-1060): the driver hands dyn_core ITS pfull (the mid-level pressure of its
    # ak, bk) and ks = 0; the profiles are evaluated on either side (libm there, numpy here), so the comparison allows rounding then
    pfull_drv = 0.5 * (ak[:-1] + ak[1:] + (bk[:-1] + bk[1:]) * 1.0e5)
    rf_cut = float(pfull_drv[npz // 2]) + 1.0
    damp = fast_tau_w_sec > 0.0 or rf_fast_tau > 0.0
    fl = DynFlags(n_split=n_split, ptop=N.PTOP, hydrostatic=hydrostatic, d_con=d_con, beta=beta, use_cond=moist, moist_kappa=moist,
                  fast_tau_w_sec=fast_tau_w_sec, rf_fast=rf_fast_tau > 0.0, tau=rf_fast_tau, rf_cutoff=rf_cut if damp else 30.0e2,
                  fill_dp=fill_dp)
    mo = None
    if moist:     # q_con / cappa as moist_cv gives them for small mixing ratios of six species (halos periodic: the caller's, :464-465)
        import parity_remap as R
        from gfdl_atmos_cubed_sphere_amd.layout import periodic_fill
        from gfdl_atmos_cubed_sphere_amd.lib import CP_AIR, RDGAS
        rng = np.random.default_rng(5)
        qq = rng.uniform(0.0, 1.0, bd.shape("A", npz) + (6,)) * np.array([0.02] + [0.002] * 5)
        cvm, qc = N.np_moist_cv(qq, dict(R.MOIST6, sphum=1), CP_AIR - RDGAS)
        mo = dict(use_cond=True, moist_kappa=True, q_con=np.asfortranarray(qc), cappa=np.asfortranarray(RDGAS / (RDGAS + cvm / (1.0 + 0.6077 * qq[..., 0]))))
        for a_ in (mo["q_con"], mo["cappa"]):
            for k in range(npz):
                periodic_fill(bd, a_[:, :, k], "A")
    dp_ref = (ak[1:] - ak[:-1]) + (bk[1:] - bk[:-1]) * 1.0e5
    ctx = Context(g, npz, lib=lib)
    try:
        dc = DynCore(ctx, fl, dp_ref, pfull=pfull_drv if damp else None, ks=0, akbk=(ak, bk) if fill_dp else None)
        dc.set_state(st["u"], st["v"], st["w"], st["delp"], st["pt"], st["delz"], st["phis"])
        if mo:
            dc.d["q_con"].upload(mo["q_con"])
            dc.d["cappa"].upload(mo["cappa"])
        for _ in range(nsteps):
            if mo:          # the halo updates fv_dynamics makes before every dyn_core call (:464-465); the driver's arrays keep theirs
                dc.halo.update([(dc.d["q_con"], "A")])
            dc.run(bdt)
        names = ("u", "v", "delp", "pt", "mfx", "cx") + (() if hydrostatic else ("w", "delz")) + (("q_con",) if mo else ())
        ref = {n: dc.d[n].download() for n in names}
        if "pkz" in dc.d and (hydrostatic or d_con > 1e-5):
            ref["pkz"] = dc.d["pkz"].download()
        if do_diss_est:
            ref["diss_est"] = dc.d["diss_est"].download()
            assert np.max(np.abs(ref["diss_est"])) > 0.0
    finally:
        ctx.close()
    exe = build_refsig(workdir, libdir=os.path.dirname(lib.path), libname=os.path.basename(lib.path)[3:-3])
    fin, fout = os.path.join(str(workdir), "in_rs.bin"), os.path.join(str(workdir), "out_rs.bin")
    write_input(fin, bd, npz, 0, n_split, 1, nsteps, False, 1000.0, 1000.0, float(g.m["f0"][0, 0]), bdt, N.PTOP, ak, bk, st, None,
                hydrostatic=hydrostatic, d_con=d_con, d_ext=fl.d_ext, beta=beta, moist=mo)
    spec = [(n, k, ()) for n, k in (("u", "U"), ("v", "V"), ("w", "A"), ("delp", "A"), ("pt", "A"), ("delz", "CC"), ("mfx", "FX"), ("cx", "CX"),
                                    ("pkz", "CC")) + ((("q_con", "A"),) if mo else ()) + ((("diss_est", "A"),) if do_diss_est else ())]
    env = {}
    if do_diss_est:
        env["FV3_REFSIG_DISS_EST"] = "1"
    if fill_dp:
        env["FV3_REFSIG_FILL_DP"] = "1"
    if fast_tau_w_sec > 0.0:
        env["FV3_REFSIG_FAST_TAU_W"] = repr(float(fast_tau_w_sec))
    if rf_fast_tau > 0.0:
        env["FV3_REFSIG_RF_FAST"] = repr(float(rf_fast_tau))
    if damp:
        env["FV3_REFSIG_RF_CUTOFF"] = repr(rf_cut)
    if registry:   # the lazy host-address registry: arrays the driver never writes are copied in once, results fetched once at the end
        env["FV3_REFSIG_REGISTRY"] = "1"
    os.environ.update(env)
    try:
        res, out = _run_refsig(lib, exe, fin, fout, "", layout, bd, npz, spec)
    finally:
        for k in env:
            os.environ.pop(k, None)
    _compare_blocks(res, ref, bd, "reference-signature dyn_core", tol=1e-13 if damp else None)
    import re
    m = re.search(r"registry \(h2d copies, h2d skipped, d2h copies, d2h deferred\) (\d+) (\d+) (\d+) (\d+)", out)
    assert m, out[-500:]
    h2d, h2d_skip, d2h, d2h_def = (int(x) for x in m.groups())
    n_in = 12 + (1 if do_diss_est else 0)       # arrays dyn_core is handed through the registry (diss_est rides along with do_diss_est)
    if registry:
        assert h2d == n_in and h2d_skip == n_in * (nsteps - 1) and d2h_def > 0 and d2h == d2h_def // nsteps, (h2d, h2d_skip, d2h, d2h_def)
    else:
        assert h2d == n_in * nsteps and h2d_skip == 0 and d2h_def == 0 and d2h > 0, (h2d, h2d_skip, d2h, d2h_def)
    if fill_dp:   # mix_dp is in the run at all
        ctx = Context(g, npz, lib=lib)
        try:
            dc = DynCore(ctx, DynFlags(n_split=n_split, ptop=N.PTOP, hydrostatic=hydrostatic, d_con=d_con, beta=beta), dp_ref)
            dc.set_state(st["u"], st["v"], st["w"], st["delp"], st["pt"], st["delz"], st["phis"])
            for _ in range(nsteps):
                dc.run(bdt)
            assert P.rel_rms(dc.d["delp"].download(), ref["delp"]) > 1e-6
        finally:
            ctx.close()
    if damp:   # the damping is in the run at all
        ctx = Context(g, npz, lib=lib)
        try:
            dc = DynCore(ctx, DynFlags(n_split=n_split, ptop=N.PTOP, hydrostatic=hydrostatic, d_con=d_con, beta=beta), dp_ref)
            dc.set_state(st["u"], st["v"], st["w"], st["delp"], st["pt"], st["delz"], st["phis"])
            for _ in range(nsteps):
                dc.run(bdt)
            assert P.rel_rms(dc.d["u"].download(), ref["u"]) > 1e-7
        finally:
            ctx.close()
    return out


def _run_refsig(lib, exe, fin, fout, mode, layout, bd, npz, spec):
    """run the driver on px x py processes (or one) and read what every rank wrote: [(block Bounds, {name: array})]"""
    import ctypes as C
    from gfdl_atmos_cubed_sphere_amd.layout import Bounds
    px, py = layout
    nranks = px * py
    args = [exe, fin, fout] + ([mode] if mode or nranks > 1 else [])
    if nranks == 1:
        r = subprocess.run(args, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs, files, blocks = [r.stdout], [fout], [bd]
    else:
        if not mode:
            args[3] = "dyn_core"
        uid = (C.c_ubyte * 128)()
        lib.check(lib.dll.fv3_comm_get_unique_id(uid), "fv3_comm_get_unique_id")
        idf = fin + ".id"
        with open(idf, "wb") as f:
            f.write(bytes(uid))
        procs = [subprocess.Popen(args + [str(rk), str(nranks), str(px), str(py), idf], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
                 for rk in range(nranks)]
        outs = []
        for p_ in procs:
            o, _ = p_.communicate(timeout=1500)
            outs.append(o)
            assert p_.returncode == 0, o[-3000:]
        files = [fout + f".{rk}" for rk in range(nranks)]
        bnx, bny = bd.nx // px, bd.ny // py
        blocks = [Bounds(1 + (rk % px) * bnx, (rk % px + 1) * bnx, 1 + (rk // px) * bny, (rk // px + 1) * bny) for rk in range(nranks)]
    res = []
    for fn, b in zip(files, blocks):
        got = {}
        with open(fn, "rb") as f:
            for n, kind, extra in spec:
                shp = b.shape(kind, npz) + extra
                got[n] = np.fromfile(f, dtype=np.float64, count=int(np.prod(shp))).reshape(shp, order="F")
        res.append((b, got))
    return res, outs[0]


def _compare_blocks(res, ref, bd, what, tol=None):
    """every rank's block against the same block of the single-domain reference (compute domain of every field kind)"""
    kinds = {"u": "U", "v": "V", "w": "A", "delp": "A", "pt": "A", "q_con": "A", "ua": "A", "diss_est": "A", "delz": "CC", "mfx": "FX", "cx": "CX", "pkz": "CC", "q": "A"}
    for b, got in res:
        for n in ref:
            kind = kinds[n]
            lim = dict(A=(b.is_, b.ie, b.js, b.je), U=(b.is_, b.ie, b.js, b.je + 1), V=(b.is_, b.ie + 1, b.js, b.je),
                       CC=(b.is_, b.ie, b.js, b.je), FX=(b.is_, b.ie + 1, b.js, b.je), CX=(b.is_, b.ie + 1, b.js, b.je))[kind]
            a = b.view(got[n], kind, *lim) if kind in ("A", "U", "V") else got[n] if kind != "CX" else got[n][:, b.ng:b.ng + b.ny]
            if n == "q":
                a = got[n][b.ng:b.ng + b.nx, b.ng:b.ng + b.ny]
                r_ = ref[n][bd.ng + b.is_ - 1:bd.ng + b.ie, bd.ng + b.js - 1:bd.ng + b.je]
            elif kind in ("A", "U", "V"):
                r_ = bd.view(ref[n], kind, *lim)
            elif kind == "CC":
                r_ = ref[n][b.is_ - 1:b.ie, b.js - 1:b.je]
            elif kind == "FX":
                r_ = ref[n][b.is_ - 1:b.ie + 1, b.js - 1:b.je]
            else:   # CX: (nx + 1, njd): rows jsd .. jed of the global array
                r_ = ref[n][b.is_ - 1:b.ie + 1, bd.ng + b.js - 1:bd.ng + b.je]
            assert np.all(np.isfinite(a)), n
            if tol is not None:
                assert np.max(np.abs(a - r_)) <= tol * max(np.max(np.abs(r_)), 1e-300), f"{n}: {what} and the Python host differ beyond {tol} (max abs {np.max(np.abs(a - r_)):.3e})"
                continue
            assert np.array_equal(a, r_), f"{n} (block {b.is_}:{b.ie}, {b.js}:{b.je}): {what} and the Python host differ (max abs {np.max(np.abs(a - r_)):.3e})"


def check_fortran_fv_dynamics(lib, workdir, nx=24, ny=16, npz=8, nq=2, n_split=2, k_split=2, nsteps=2, bdt=8.0, hydrostatic=False, layout=(1, 1),
                              consv_te=0.0, tau=0.0, moist=False, do_diss_est=False, consv_am=False, beta=0.0, hybrid_z=False):
    """fv_dynamics called with the reference's argument list on host arrays (fv3_dyn_core_mod::fv_dynamics, driver fv3_solo_refsig in
    its fv_dynamics mode: T -> theta_v, k_split x (dyn_core, tracer_2d, remap), last_step, cubed_to_latlon) against the Python host's
    FvDynamics.step_from_temperature on the same state: u, v, w, delp, pt (T), delz, the tracers and ua bit-identical"""
    import parity_common as P
    import parity_dyn as D
    import parity_nh as N
    from gfdl_atmos_cubed_sphere_amd.dyn_core import DynFlags
    from gfdl_atmos_cubed_sphere_amd.fv_dynamics import FvDynamics
    from gfdl_atmos_cubed_sphere_amd.layout import Bounds
    from gfdl_atmos_cubed_sphere_amd.lib import Context
    bd = Bounds(1, nx, 1, ny)
    g = P.make_grid(bd, False)
    if do_diss_est:
        import dataclasses
        g = dataclasses.replace(g, do_diss_est=True, prevent_diss_cooling=False)
    st, _ = D.make_state(bd, npz)
    sig = np.linspace(0.0, 1.0, npz + 1) ** 1.5
    ak, bk = N.PTOP * (1.0 - sig), sig.copy()
    rng = np.random.default_rng(5)
    q = np.asfortranarray(rng.uniform(0.0, 1.0, bd.shape("A", npz) + (nq,))) if nq else None
    mo = None
    if moist:      # thermostruct%use_cond = moist_kappa = .true. (the reference's defaults, fv_arrays.F90:1226-1227): six water species
        import parity_remap as R
        assert nq >= 6 and not hydrostatic
        q[..., :6] *= 1.0e-3                       # small mixing ratios
        mo = dict(use_cond=True, moist_kappa=True, q_con=bd.zeros("A", npz), cappa=bd.zeros("A", npz))   # fv_dynamics forms both itself
    fl = DynFlags(n_split=n_split, ptop=N.PTOP, hydrostatic=hydrostatic, use_cond=moist, moist_kappa=moist, beta=beta)
    ca = None
    if consv_am:   # flagstruct%consv_am with a made-up grid (parity_dyn.check_fv_cycle_from_temperature): latitudes, l2c_u / l2c_v (zero outside
                   # the compute domain, as the reference's members end there) and zxg of no sphere
        rng_ = np.random.default_rng(31)
        lat = np.asfortranarray(np.arccos(rng_.uniform(0.2, 1.0, bd.shape("A"))))
        lu, lv = bd.zeros("U"), bd.zeros("V")
        ng = bd.ng
        lu[ng:ng + nx, ng:ng + ny + 1] = rng_.uniform(-1, 1, (nx, ny + 1))
        lv[ng:ng + nx + 1, ng:ng + ny] = rng_.uniform(-1, 1, (nx + 1, ny))
        ca = dict(lat=lat, coslat=np.asfortranarray(np.cos(lat)), l2c_u=lu, l2c_v=lv, zxg=rng_.uniform(-1e-3, 1e-3, (nx, ny)), omega=7.292e-5)
    ctx = Context(g, npz, lib=lib)
    try:
        fv = FvDynamics(ctx, fl, ak, bk, nq=nq, k_split=k_split, adiabatic=not moist, c2l_ord=4, consv_te=consv_te, tau=tau, moist_phys=False,
                        moist=dict(R.MOIST6) if moist else None, consv_am=ca)
        fv.dc.set_state(st["u"], st["v"], st["w"], st["delp"], st["pt"], st["delz"], st["phis"])
        if nq:
            fv.set_tracers(q)
        if hydrostatic:       # the hydrostatic conversion takes pkz as the state holds it: 1 here (the file's pt is theta), as in the driver
            fv.dc.d["pkz"].upload(np.asfortranarray(np.ones(bd.shape("CC", npz))))
        for _ in range(nsteps):
            fv.step_from_temperature(bdt)
        d = fv.dc.d
        ref = {n: d[n].download() for n in (("u", "v", "delp", "pt", "ua") if hydrostatic else ("u", "v", "w", "delp", "pt", "delz", "ua"))}
        if nq:
            ref["q"] = d["q"].download()
        if moist:
            ref["q_con"] = d["q_con"].download()
        if do_diss_est:      # zeroed at the first cycle of every call, summed over all k_split x n_split substeps of it
            ref["diss_est"] = d["diss_est"].download()
            assert np.max(np.abs(ref["diss_est"])) > 0.0
    finally:
        ctx.close()
    exe = build_refsig(workdir, libdir=os.path.dirname(lib.path), libname=os.path.basename(lib.path)[3:-3])
    fin, fout = os.path.join(str(workdir), "in_fd.bin"), os.path.join(str(workdir), "out_fd.bin")
    write_input(fin, bd, npz, nq, n_split, k_split, nsteps, True, 1000.0, 1000.0, float(g.m["f0"][0, 0]), bdt, N.PTOP, ak, bk, st, q,
                hydrostatic=hydrostatic, d_con=0.0, d_ext=fl.d_ext, beta=beta, moist=mo, consv_am=ca)
    spec = [(n, k, ()) for n, k in (("u", "U"), ("v", "V"), ("w", "A"), ("delp", "A"), ("pt", "A"), ("delz", "CC"))]
    if nq:
        spec.append(("q", "A", (nq,)))
    spec.append(("ua", "A", ()))
    if moist:
        spec.append(("q_con", "A", ()))
    if do_diss_est:
        spec.append(("diss_est", "A", ()))
        os.environ["FV3_REFSIG_DISS_EST"] = "1"
    os.environ["FV3_SOLO_CONSV_TE"], os.environ["FV3_SOLO_TAU"] = repr(float(consv_te)), repr(float(tau))
    if hybrid_z:    # handed on to Lagrangian_to_Eulerian and never read there (fv_mapz.F90:62, :128): the same results
        os.environ["FV3_REFSIG_HYBRID_Z"] = "1"
    try:
        res, out = _run_refsig(lib, exe, fin, fout, "fv_dynamics", layout, bd, npz, spec)
    finally:
        os.environ.pop("FV3_SOLO_CONSV_TE", None)
        os.environ.pop("FV3_SOLO_TAU", None)
        os.environ.pop("FV3_REFSIG_DISS_EST", None)
        os.environ.pop("FV3_REFSIG_HYBRID_Z", None)
    # consv_am: u00 is a difference of column integrals ~ r^2 omega dm, which amplifies what cos() of the two run-time libraries differs by
    _compare_blocks(res, ref, bd, "reference-signature fv_dynamics", tol=1e-11 if consv_am else None)
    if consv_am:
        assert abs(fv.last_u00) > 1e-8
    return out


def build_solo_sphere(workdir, libdir=CSRC, libname="fv3_mi355x"):
    return _compile(workdir, "fv3_solo_sphere", ('fv3_mi355x_mod.F90', 'fv3_host_mod.F90', 'fv3_sphere_mod.F90', 'fv3_solo_sphere.F90'), libdir, libname)


_GH_A = ["area", "rarea", "dxa", "dya", "rdxa", "rdya", "cosa_s", "rsin2", "f0"]
_GH_U = ["dx", "rdx", "dyc", "rdyc", "cosa_v", "sina_v", "rsin_v", "divg_u", "del6_u"]
_GH_V = ["dy", "rdy", "dxc", "rdxc", "cosa_u", "sina_u", "rsin_u", "divg_v", "del6_v"]
_GH_B = ["rarea_c", "fC", "cosa", "sina"]


def check_fortran_sphere(lib, workdir, npx=13, npz=20, nq=2, n_split=2, k_split=2, nsteps=1, bdt=900.0, hydrostatic=False, d_con=0.0, beta=0.0,
                         inline_q=False, remap_te=False):
    """the Jablonowski-Williamson state on the six faces through (a) the Python host (FvDynamics over MultiContext, device-gather halo
    updates) and (b) the Fortran host (fv3_sphere_mod: one context per face, every halo update through the cube-edge exchange behind
    the C ABI, mpp_get_boundary after the last substep, adv_pe): bit-identical states on every face"""
    import cubed_common as CC
    import parity_cubed as PC
    from gfdl_atmos_cubed_sphere_amd import lib as L
    from gfdl_atmos_cubed_sphere_amd.cubed_dyn import CubeHaloAdapter, MultiContext
    from gfdl_atmos_cubed_sphere_amd.dyn_core import DynFlags
    from gfdl_atmos_cubed_sphere_amd.fv_dynamics import FvDynamics
    from gfdl_atmos_cubed_sphere_amd.lib import Context
    cs, gs = CC.sphere(npx)
    nx = npx - 1
    sig = np.linspace(0.0, 1.0, npz + 1) ** 1.5
    ak, bk = 300.0 * (1.0 - sig), sig.copy()
    st = cs.jablonowski_williamson(ak, bk, hydrostatic=hydrostatic, rdgas=L.RDGAS, grav=L.GRAV)
    CC.exchange(cs, st, ("phis",), "A")
    fl = DynFlags(n_split=n_split, hydrostatic=hydrostatic, ptop=float(ak[0]), d_con=d_con, beta=beta, inline_q=inline_q, **(dict(d_ext=0.0) if hydrostatic else {}))
    bd = gs[0].bd
    ng = bd.ng
    c = (slice(ng, ng + nx), slice(ng, ng + nx))
    for s_ in st:          # T -> theta (the host's job before dyn_core)
        if hydrostatic:
            pe = ak[0] + np.concatenate([np.zeros(s_["delp"].shape[:2] + (1,)), np.cumsum(s_["delp"], axis=2)], axis=2)[c]
            peln = np.log(pe)
            pkz = (pe[:, :, 1:] ** fl.akap - pe[:, :, :-1] ** fl.akap) / (fl.akap * (peln[:, :, 1:] - peln[:, :, :-1]))
            s_["w"] = np.zeros_like(s_["delp"])
            s_["delz"] = bd.zeros("CC", npz)
        else:
            pkz = ((-fl.rdgas / fl.grav) * s_["delp"][c] * s_["pt"][c] / s_["delz"]) ** fl.akap
        s_["pt"][c] = s_["pt"][c] / pkz
    q = PC.tracer_fields(cs, npz, nq) if nq else None
    # ---- (a) Python host ----
    mctx = MultiContext([Context(g, npz, lib=lib) for g in gs])
    try:
        fv = FvDynamics(mctx, fl, ak, bk, nq=nq, k_split=k_split, halo=CubeHaloAdapter(mctx, npx, topo=CC.product_topo(npx)), remap_te=remap_te)
        fv.dc.set_state([s_["u"] for s_ in st], [s_["v"] for s_ in st], [s_["w"] for s_ in st], [s_["delp"] for s_ in st],
                        [s_["pt"] for s_ in st], [s_["delz"] for s_ in st], [s_["phis"] for s_ in st])
        if nq:
            fv.set_tracers(q)
        for _ in range(nsteps):
            fv.step(bdt)
        d = fv.dc.d
        names = ("u", "v", "delp", "pt") if hydrostatic else ("u", "v", "w", "delp", "pt", "delz")
        ref = {n: d[n].download() for n in names}
        if nq:
            ref["q"] = d["q"].download()
    finally:
        mctx.close()
    # ---- (b) Fortran host ----
    exe = build_solo_sphere(workdir, libdir=os.path.dirname(lib.path), libname=os.path.basename(lib.path)[3:-3])
    fin, fout = os.path.join(str(workdir), "sph_in.bin"), os.path.join(str(workdir), "sph_out.bin")
    F = lambda a: np.asfortranarray(a, dtype=np.float64).ravel(order="F")      # noqa: E731
    with open(fin, "wb") as f:
        np.array([npx, npz, nq, n_split, k_split, nsteps, 0, int(hydrostatic) + 2 * int(inline_q) + 4 * int(remap_te), fl.nord], dtype=np.int32).tofile(f)
        np.array([bdt, fl.ptop, d_con, fl.d_ext, gs[0].da_min, gs[0].da_min_c, fl.d4_bg, fl.beta], dtype=np.float64).tofile(f)
        np.asarray(ak, dtype=np.float64).tofile(f)
        np.asarray(bk, dtype=np.float64).tofile(f)
        for t in range(6):
            m = gs[t].m
            for grp in (_GH_A, _GH_U, _GH_V, _GH_B):
                for n in grp:
                    F(m[n]).tofile(f)
            F(m["sin_sg"]).tofile(f); F(m["cos_sg"]).tofile(f)
            for n in ("edge_w", "edge_e", "edge_s", "edge_n"):
                np.asarray(m[n], dtype=np.float64).tofile(f)
            F(m["rsina"]).tofile(f)
            np.asarray(m["corner_f"], dtype=np.float64).ravel().tofile(f)
            for n in ("a11", "a12", "a21", "a22", "ec1", "ec2", "en1", "en2"):
                F(m[n]).tofile(f)
            for n in ("u", "v", "w", "delp", "pt", "delz", "phis"):
                F(st[t][n]).tofile(f)
            if nq:
                F(q[t]).tofile(f)
    r = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "fv3_cube_halo_start" in r.stdout, r.stdout
    i0, i1, j0, j1 = bd.is_, bd.ie, bd.js, bd.je
    rng_ = {"u": ("U", i0, i1, j0, j1 + 1), "v": ("V", i0, i1 + 1, j0, j1), "w": ("A", i0, i1, j0, j1),
            "delp": ("A", i0, i1, j0, j1), "pt": ("A", i0, i1, j0, j1)}
    with open(fout, "rb") as f:
        for t in range(6):
            got = {}
            for n, kind in (("u", "U"), ("v", "V"), ("w", "A"), ("delp", "A"), ("pt", "A"), ("delz", "CC")):
                shp = bd.shape(kind, npz)
                got[n] = np.fromfile(f, dtype=np.float64, count=int(np.prod(shp))).reshape(shp, order="F")
            if nq:
                shp = bd.shape("A", npz) + (nq,)
                got["q"] = np.fromfile(f, dtype=np.float64, count=int(np.prod(shp))).reshape(shp, order="F")
            for n in ref:
                if n in rng_:
                    kind, *r4 = rng_[n]
                    a, b = bd.view(got[n], kind, *r4), bd.view(ref[n][t], kind, *r4)
                elif n == "q":
                    a, b = got[n][c], ref[n][t][c]
                else:
                    a, b = got[n], ref[n][t]
                assert np.all(np.isfinite(a)), (t, n)
                assert np.array_equal(a, b), f"face {t + 1} {n}: Fortran host and Python host differ (max abs {np.max(np.abs(a - b)):.3e})"
    return r.stdout


def build_solo_refsig_sphere(workdir, libdir=CSRC, libname="fv3_mi355x"):
    return _compile(workdir, "fv3_solo_refsig_sphere", ('fv3_mi355x_mod.F90', 'fv3_host_mod.F90', 'fv3_sphere_mod.F90', 'fv3_dyn_core_mod.F90', 'fv3_solo_refsig_sphere.F90'), libdir, libname)


def check_refsig_sphere(lib, workdir, npx=13, npz=20, nq=2, n_split=2, k_split=2, bdt=900.0, hydrostatic=False, consv_te=1.0, tau=10.0,
                        zvir=0.6077, face_rank=(0, 0, 0, 0, 0, 0), have_grid=False, tol=0.0, what="fv_dynamics", thermo=False,
                        do_diss_est=False, fill_dp=False, consv_am=False, beta=0.0):
    """fv_dynamics WITH THE REFERENCE'S ARGUMENT LIST on the cubed sphere (fv3_dyn_core_mod.F90: one call per tile, host arrays with the
    fv_arrays layout, gridstruct / flagstruct / bd / domain) against the Python host's whole fv_dynamics call
    (FvDynamics.step_from_temperature over the six contexts): compute_total_energy, T -> theta_v with the virtual effect, Rayleigh_Super
    (tau > 0), the k_split loop, the energy fixer (consv_te) and cubed_to_latlon -- identical bits on every tile (tol = 0), with the
    tiles in one process or spread over processes (face_rank; the exchange then runs between them).  have_grid: corner_f is not
    handed over, the wrapper forms it from grid / agrid as the reference's a2b_ord4 does (then held to `tol`)."""
    import ctypes as C
    import cubed_common as CC
    import parity_cubed as PC
    from gfdl_atmos_cubed_sphere_amd import lib as L
    from gfdl_atmos_cubed_sphere_amd.cubed_dyn import CubeHaloAdapter, MultiContext
    from gfdl_atmos_cubed_sphere_amd.dyn_core import DynFlags
    from gfdl_atmos_cubed_sphere_amd.fv_dynamics import FvDynamics
    from gfdl_atmos_cubed_sphere_amd.lib import Context
    if do_diss_est:     # flagstruct%do_diss_est is a member of the gridstruct the contexts upload; the sphere's grids are shared: set, run, restore
        _, gs_ = CC.sphere(npx)
        old = [(g.do_diss_est, g.prevent_diss_cooling) for g in gs_]
        for g in gs_:
            g.do_diss_est, g.prevent_diss_cooling = True, False
        try:
            return check_refsig_sphere(lib, workdir, npx, npz, nq, n_split, k_split, bdt, hydrostatic, consv_te, tau, zvir, face_rank, have_grid, tol,
                                       what, thermo, False, fill_dp, consv_am, beta)
        finally:
            for g, o in zip(gs_, old):
                g.do_diss_est, g.prevent_diss_cooling = o
    cs, gs = CC.sphere(npx)
    do_diss_est = bool(gs[0].do_diss_est)
    nx = npx - 1
    sig = np.linspace(0.0, 1.0, npz + 1) ** 1.5
    ak, bk = 300.0 * (1.0 - sig), sig.copy()
    st = cs.jablonowski_williamson(ak, bk, hydrostatic=hydrostatic, rdgas=L.RDGAS, grav=L.GRAV)      # pt = T
    CC.exchange(cs, st, ("phis",), "A")
    # thermo: thermostruct%use_cond = moist_kappa = .true. (the reference's defaults, fv_arrays.F90:1226-1227) -- six water species; fv_dynamics forms
    # q_con / cappa itself (moist_cv), dyn_core is handed them
    assert not (thermo and hydrostatic)
    if thermo and what == "fv_dynamics":
        nq = max(nq, 6)
    fl = DynFlags(n_split=n_split, hydrostatic=hydrostatic, ptop=float(ak[0]), use_cond=thermo, moist_kappa=thermo, fill_dp=fill_dp, beta=beta,
                  **(dict(d_ext=0.0) if hydrostatic else {}))
    bd = gs[0].bd
    ng = bd.ng
    c = (slice(ng, ng + nx), slice(ng, ng + nx))
    for s_ in st:
        if hydrostatic:
            s_["w"] = np.zeros_like(s_["delp"])
            s_["delz"] = bd.zeros("CC", npz)
    ak_call, bk_call = ak, bk
    if fill_dp:     # flagstruct%fill_dp: the ak / bk dyn_core is handed have reference thicknesses in the middle of each layer's range on the
        # sphere -- mix_dp acts on about half of the cells (the state itself keeps the levels it was built on); one dyn_core call only
        # (fv_dynamics remaps to the levels of its ak / bk)
        assert what == "dyn_core"
        lo = np.min([s_["delp"][c].min(axis=(0, 1)) for s_ in st], axis=0)
        hi = np.max([s_["delp"][c].max(axis=(0, 1)) for s_ in st], axis=0)
        ak_call, bk_call = np.zeros(npz + 1), np.concatenate(([0.0], np.cumsum(100.0 * 0.5 * (lo + hi)))) / 1.0e5
        ak_call[0] = 1.0          # (a positive model top)
        dpmin = 0.01 * ((ak_call[1:] - ak_call[:-1]) + (bk_call[1:] - bk_call[:-1]) * 1.0e5)
        assert sum(int(np.sum(s_["delp"][c] < dpmin)) for s_ in st) > 0.2 * 6 * nx * nx * npz      # mix_dp has work to do
    # what p_var (fv_grid_utils / init_case) leaves of the hydrostatic pressures: pe (is-1:ie+1, npz+1, js-1:je+1), pk, peln, pkz
    pv = []
    for s_ in st:
        e = (slice(ng - 1, ng + nx + 1), slice(ng - 1, ng + nx + 1))
        if hydrostatic:
            pe_ = ak[0] + np.concatenate([np.zeros(s_["delp"].shape[:2] + (1,)), np.cumsum(s_["delp"], axis=2)], axis=2)
            pec = pe_[c]
            peln = np.log(pec)
            pk = pec ** fl.akap
            pkz = (pk[:, :, 1:] - pk[:, :, :-1]) / (fl.akap * (peln[:, :, 1:] - peln[:, :, :-1]))
            pv.append(dict(pe=np.asfortranarray(np.transpose(pe_[e], (0, 2, 1))), pk=np.asfortranarray(pk),
                           peln=np.asfortranarray(np.transpose(peln, (0, 2, 1))), pkz=np.asfortranarray(pkz)))
        else:
            pv.append(dict(pe=np.zeros((nx + 2, npz + 1, nx + 2), order="F"), pk=np.zeros((nx, nx, npz + 1), order="F"),
                           peln=np.zeros((nx, npz + 1, nx), order="F"), pkz=np.zeros((nx, nx, npz), order="F")))
    if what == "dyn_core":                  # one dyn_core call (model/dyn_core.F90:94-98): pt is theta_v on entry, no tracers
        nq, consv_te, tau = 0, 0.0, 0.0
        for s_, p_ in zip(st, pv):
            pkz = p_["pkz"] if hydrostatic else ((-fl.rdgas / fl.grav) * s_["delp"][c] * s_["pt"][c] / s_["delz"]) ** fl.akap
            s_["pt"][c] = s_["pt"][c] / pkz
            if not hydrostatic:
                p_["pkz"] = np.asfortranarray(pkz)
    ca = None
    if consv_am:    # flagstruct%consv_am (fv_dynamics.F90:358-361, :747-800): the tiles' own latitudes (gridstruct%agrid: have_grid), made-up
        # l2c_u / l2c_v (the reference's are projections of the east vector; any field exercises the correction) and mountain torque zxg
        assert have_grid and what == "fv_dynamics"
        rng = np.random.default_rng(77)
        ca = dict(coslat=[np.asfortranarray(np.cos(g.m["agrid"][:, :, 1])) for g in gs],
                  l2c_u=[bd.zeros("U") for _ in range(6)], l2c_v=[bd.zeros("V") for _ in range(6)],
                  zxg=[np.asfortranarray(1.0e-3 * rng.uniform(-1, 1, (nx, nx))) for _ in range(6)])
        for t in range(6):
            ca["l2c_u"][t][ng:ng + nx, ng:ng + nx + 1] = rng.uniform(0.2, 1.0, (nx, nx + 1))
            ca["l2c_v"][t][ng:ng + nx + 1, ng:ng + nx] = rng.uniform(-0.5, 0.5, (nx + 1, nx))
    q = PC.tracer_fields(cs, npz, nq) if nq else None
    if nq:                                  # the first tracer is the specific humidity of the virtual effect: small and positive
        for t in range(6):
            q[t][..., 0] = 0.01 * np.abs(q[t][..., 0]) / (1.0e-30 + np.abs(q[t][..., 0]).max())
    moist = bool(nq) and zvir > 0.0
    qc_in = cp_in = None
    if thermo and nq:
        for t in range(6):
            q[t][..., 1:6] = 1.0e-3 * np.abs(q[t][..., 1:6]) / (1.0e-30 + np.abs(q[t][..., 1:6]).max())   # small condensate mixing ratios
    if thermo and what == "dyn_core":
        rng = np.random.default_rng(41)
        qc_in = [np.asfortranarray(rng.uniform(0.0, 0.01, bd.shape("A", npz))) for _ in range(6)]
        cp_in = [np.asfortranarray(0.28 + rng.uniform(0.0, 0.005, bd.shape("A", npz))) for _ in range(6)]
    # ---- (a) Python host ----
    mctx = MultiContext([Context(g, npz, lib=lib) for g in gs])
    try:
        import parity_remap as R
        fv = FvDynamics(mctx, fl, ak_call, bk_call, nq=nq, k_split=k_split, halo=CubeHaloAdapter(mctx, npx, topo=CC.product_topo(npx)),
                        consv_te=consv_te, tau=tau, adiabatic=not moist, moist_phys=False, moist=dict(R.MOIST6) if thermo else None,
                        consv_am=ca)
        fv.remap_par["r_vir"] = zvir if moist else fv.remap_par["r_vir"]
        fv.dc.set_state([s_["u"] for s_ in st], [s_["v"] for s_ in st], [s_["w"] for s_ in st], [s_["delp"] for s_ in st],
                        [s_["pt"] for s_ in st], [s_["delz"] for s_ in st], [s_["phis"] for s_ in st])
        if nq:
            fv.set_tracers(q)
        for n in ("pe", "pk", "peln", "pkz"):
            fv.dc.d[n].upload([p_[n] for p_ in pv])
        if what == "dyn_core":
            if thermo:      # the halo updates fv_dynamics makes in front of dyn_core (:464-465)
                fv.dc.d["q_con"].upload(qc_in)
                fv.dc.d["cappa"].upload(cp_in)
                fv.dc.halo.update([(fv.dc.d["q_con"], "A")])
                fv.dc.halo.update([(fv.dc.d["cappa"], "A")])
            fv.dc.run(bdt, end_step=True)
        else:
            fv.step_from_temperature(bdt)
        d = fv.dc.d
        names = ("u", "v", "delp", "pt", "ua", "va") if hydrostatic else ("u", "v", "w", "delp", "pt", "delz", "ua", "va")
        if what == "dyn_core":
            names = tuple(n for n in names if n not in ("ua", "va")) + ("mfx", "cx")
        ref = {n: d[n].download() for n in names}
        if nq:
            ref["q"] = d["q"].download()
        if thermo:
            ref["q_con"] = d["q_con"].download()
            assert max(float(np.max(np.abs(x[c]))) for x in ref["q_con"]) > 0.0
        if do_diss_est:
            ref["diss_est"] = d["diss_est"].download()
            assert max(float(np.max(np.abs(x[c]))) for x in ref["diss_est"]) > 0.0
        if consv_am:
            assert abs(fv.last_u00) > 1e-8, fv.last_u00           # the correction is in the run
        for n in names:
            assert all(np.all(np.isfinite(x[c])) for x in ref[n]), f"the Python host's {n} is not finite"
        assert tau <= 0.0 or fv._rf[2] > 0, "the Rayleigh damping acts on no level of this test"
    finally:
        mctx.close()
    # ---- (b) the reference's argument list ----
    exe = build_solo_refsig_sphere(workdir, libdir=os.path.dirname(lib.path), libname=os.path.basename(lib.path)[3:-3])
    nranks = max(face_rank) + 1
    uid = (C.c_ubyte * 128)()
    if nranks > 1:
        lib.check(lib.dll.fv3_comm_get_unique_id(uid), "fv3_comm_get_unique_id")
    F = lambda a: np.asfortranarray(a, dtype=np.float64).ravel(order="F")      # noqa: E731
    fout = os.path.join(str(workdir), "rs_out.bin")
    procs = []
    for rank in range(nranks):
        fin = os.path.join(str(workdir), f"rs_in_{rank}.bin")
        with open(fin, "wb") as f:
            np.array([npx, npz, nq, n_split, k_split, int(hydrostatic) + 8 * int(thermo) + 16 * int(do_diss_est) + 32 * int(fill_dp) + 64 * int(consv_am), fl.nord, rank,
                      nranks, int(have_grid)] + list(face_rank), dtype=np.int32).tofile(f)
            np.array([bdt, fl.ptop, 0.0, fl.d_ext, gs[0].da_min, gs[0].da_min_c, fl.d4_bg, fl.beta, consv_te, tau, zvir if moist else 0.0],
                     dtype=np.float64).tofile(f)
            f.write(bytes(uid))
            np.asarray(ak_call, dtype=np.float64).tofile(f)
            np.asarray(bk_call, dtype=np.float64).tofile(f)
            for t in range(6):
                m = gs[t].m
                for grp in (_GH_A, _GH_U, _GH_V, _GH_B):
                    for n in grp:
                        F(m[n]).tofile(f)
                F(m["sin_sg"]).tofile(f); F(m["cos_sg"]).tofile(f)
                for n in ("edge_w", "edge_e", "edge_s", "edge_n"):
                    np.asarray(m[n], dtype=np.float64).tofile(f)
                F(m["rsina"]).tofile(f)
                np.asarray(m["corner_f"], dtype=np.float64).ravel().tofile(f)
                for n in ("a11", "a12", "a21", "a22", "ec1", "ec2", "en1", "en2"):
                    F(m[n]).tofile(f)
                if have_grid:
                    F(m["grid"]).tofile(f); F(m["agrid"]).tofile(f)
                for n in ("u", "v", "w", "delp", "pt", "delz", "phis"):
                    F(st[t][n]).tofile(f)
                if nq:
                    F(q[t]).tofile(f)
                for n in ("pe", "pk", "peln", "pkz"):
                    F(pv[t][n]).tofile(f)
                if thermo and what == "dyn_core":
                    F(qc_in[t]).tofile(f); F(cp_in[t]).tofile(f)
                if consv_am:
                    F(ca["l2c_u"][t][ng:ng + nx, ng:ng + nx + 1]).tofile(f); F(ca["l2c_v"][t][ng:ng + nx + 1, ng:ng + nx]).tofile(f)
                    F(ca["zxg"][t]).tofile(f)
        procs.append(subprocess.Popen([exe, fin, fout, what], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        o, _ = p.communicate(timeout=1500)
        outs.append(o)
        assert p.returncode == 0, o[-3000:]
    i0, i1, j0, j1 = bd.is_, bd.ie, bd.js, bd.je
    rng_ = {"u": ("U", i0, i1, j0, j1 + 1), "v": ("V", i0, i1 + 1, j0, j1), "w": ("A", i0, i1, j0, j1), "delp": ("A", i0, i1, j0, j1),
            "pt": ("A", i0, i1, j0, j1), "ua": ("A", i0, i1, j0, j1), "va": ("A", i0, i1, j0, j1), "q_con": ("A", i0, i1, j0, j1),
            "diss_est": ("A", i0, i1, j0, j1)}
    worst = 0.0
    for rank in range(nranks):
        with open(fout + f".{rank}", "rb") as f:
            for t in range(6):
                if face_rank[t] != rank:
                    continue
                got = {}
                for n, kind in (("u", "U"), ("v", "V"), ("w", "A"), ("delp", "A"), ("pt", "A"), ("delz", "CC")):
                    shp = bd.shape(kind, npz)
                    got[n] = np.fromfile(f, dtype=np.float64, count=int(np.prod(shp))).reshape(shp, order="F")
                if nq:
                    shp = bd.shape("A", npz) + (nq,)
                    got["q"] = np.fromfile(f, dtype=np.float64, count=int(np.prod(shp))).reshape(shp, order="F")
                for n in ("ua", "va"):
                    shp = bd.shape("A", npz)
                    got[n] = np.fromfile(f, dtype=np.float64, count=int(np.prod(shp))).reshape(shp, order="F")
                for n, kind in (("mfx", "FX"), ("cx", "CX")):
                    shp = bd.shape(kind, npz)
                    got[n] = np.fromfile(f, dtype=np.float64, count=int(np.prod(shp))).reshape(shp, order="F")
                if thermo:
                    shp = bd.shape("A", npz)
                    got["q_con"] = np.fromfile(f, dtype=np.float64, count=int(np.prod(shp))).reshape(shp, order="F")
                if do_diss_est:
                    shp = bd.shape("A", npz)
                    got["diss_est"] = np.fromfile(f, dtype=np.float64, count=int(np.prod(shp))).reshape(shp, order="F")
                for n in ref:
                    if n in rng_:
                        kind, *r4 = rng_[n]
                        a, b = bd.view(got[n], kind, *r4), bd.view(ref[n][t], kind, *r4)
                    elif n == "q":
                        a, b = got[n][c], ref[n][t][c]
                    else:
                        a, b = got[n], ref[n][t]
                    assert np.all(np.isfinite(a)), (t, n)
                    if tol == 0.0:
                        assert np.array_equal(a, b), (f"tile {t + 1} {n}: fv_dynamics with the reference's argument list and the Python host differ "
                                                      f"(max abs {np.max(np.abs(a - b)):.3e}, rel rms {np.sqrt(np.mean((a - b) ** 2)) / (np.sqrt(np.mean(b ** 2)) + 1e-300):.3e})")
                    else:
                        e = float(np.sqrt(np.mean((a - b) ** 2)) / (np.sqrt(np.mean(b ** 2)) + 1e-300))
                        worst = max(worst, e)
                        assert e <= tol, (t, n, e)
    return worst
