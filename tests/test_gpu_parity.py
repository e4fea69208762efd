"""Parity gate on a real MI355X: the product library (libfv3_mi355x.so, HIP gfx950) through its
C ABI against the CPU oracle, same seeded inputs.  Tolerance: see parity_common (BASELINE.json asks
for rel-RMS < 1e-12; the kernels are built without FMA contraction and reproduce the oracle to
1e-14 or better)."""
import os

import numpy as np
import pytest

import parity_common as P
from gfdl_atmos_cubed_sphere_amd import lib as L

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def prod():
    return L.load()  # raises if the HIP library is missing: no fallback


@pytest.mark.parametrize("hord", [5, -5, 6, 7, 8, 10, 9, 11, 12, 13])
def test_fv_tp_2d_plain(prod, hord):
    P.check_fv_tp_2d(prod, hord)


@pytest.mark.parametrize("mode,nord,damp_c", [("mass_flux", -1, 0.0), ("mass_flux_damp", 1, 0.06),
                                             ("mass_flux_damp", 2, 0.06), ("plain", 0, 0.05), ("plain", 2, 0.06)])
def test_fv_tp_2d_modes(prod, mode, nord, damp_c):
    P.check_fv_tp_2d(prod, 10, mode=mode, nord=nord, damp_c=damp_c)


def test_fv_tp_2d_shapes(prod):
    P.check_fv_tp_2d(prod, 10, nx=64, ny=16)
    P.check_fv_tp_2d(prod, 8, nx=33, ny=9)
    P.check_fv_tp_2d(prod, 10, nx=7, ny=5)
    P.check_fv_tp_2d(prod, 10, nx=96, ny=96, nk=8)


@pytest.mark.parametrize("hydrostatic", [False, True])
@pytest.mark.parametrize("perturb", [False, True, "ortho"])
def test_c_sw(prod, hydrostatic, perturb):
    P.check_c_sw(prod, hydrostatic=hydrostatic, perturb=perturb)


def test_c_sw_shapes(prod):
    P.check_c_sw(prod, nx=28, ny=4, npz=2)
    P.check_c_sw(prod, nx=64, ny=16, npz=1)
    P.check_c_sw(prod, nx=61, ny=13, npz=1)
    P.check_c_sw(prod, nx=48, ny=48, npz=32, hydrostatic=True, perturb=False)   # BASELINE config 1 shape
    P.check_c_sw(prod, nx=96, ny=96, npz=16)


@pytest.mark.parametrize("hydrostatic", [False, True])
def test_d_sw_defaults(prod, hydrostatic):
    P.check_d_sw(prod, hydrostatic=hydrostatic)


def test_d_sw_cartesian_metrics(prod):
    P.check_d_sw(prod, perturb=False)


def test_d_sw_damping_heating(prod):
    P.check_d_sw(prod, par_over=dict(dddmp=0.2, kgb=1e-3),
                 lev_over=dict(nord=2, do_vort_damp=True, vtdm4=0.06, d_con=1.0, d2_bg=0.0075))


def test_d_sw_nord3_and_no_cooling_limiter(prod):
    P.check_d_sw(prod, lev_over=dict(nord=3, do_vort_damp=True, vtdm4=0.03, d_con=0.5),
                 flags=dict(prevent_diss_cooling=False, do_diss_est=True))


def test_d_sw_dcon_without_vort_damp(prod):
    P.check_d_sw(prod, lev_over=dict(nord=1, d_con=1.0))


def test_d_sw_use_cond_low_order(prod):
    P.check_d_sw(prod, use_cond=True, par_over=dict(hord_mt=6, hord_vt=6, hord_tm=5, hord_dp=-5))


def test_d_sw_shapes(prod):
    P.check_d_sw(prod, nx=33, ny=9, npz=2)
    P.check_d_sw(prod, nx=64, ny=16, npz=2)
    P.check_d_sw(prod, nx=48, ny=48, npz=32, hydrostatic=True, perturb=False)   # BASELINE config 1 shape
    P.check_d_sw(prod, nx=96, ny=96, npz=16)


def test_errors_are_loud(prod):
    from gfdl_atmos_cubed_sphere_amd.grid import doubly_periodic
    from gfdl_atmos_cubed_sphere_amd.layout import Bounds
    bd = Bounds(1, 8, 1, 8)
    g = doubly_periodic(bd, 9, 9)
    ctx = L.Context(g, 2, lib=prod)
    try:
        a = ctx.zeros("A", 2)
        with pytest.raises(L.Fv3Error):
            ctx.fv_tp_2d(a, a, a, 3, a, a, a, a)  # unsupported hord
    finally:
        ctx.close()


def test_halo_fill_periodic(prod):
    P.check_halo_periodic(prod)


# ---- nonhydrostatic column path (one shared exp / log: same bound as the stencils) ----
import parity_nh as N


def test_nh_update_dz_c(prod):
    N.check_update_dz_c(prod)
    N.check_update_dz_c(prod, nx=96, ny=96, km=16)


@pytest.mark.parametrize("a_imp", [1.0, 0.75])
def test_nh_riem_solver_c(prod, a_imp):
    N.check_riem_solver_c(prod, a_imp=a_imp)
    N.check_riem_solver_c(prod, nx=96, ny=64, km=32, a_imp=a_imp)


@pytest.mark.parametrize("a_imp,use_logp,last_call,fp_out", [(1.0, False, True, False), (0.75, False, True, False),
                                                            (1.0, True, False, True), (0.75, True, True, True)])
def test_nh_riem_solver3(prod, a_imp, use_logp, last_call, fp_out):
    N.check_riem_solver3(prod, a_imp=a_imp, use_logp=use_logp, last_call=last_call, fp_out=fp_out)
    N.check_riem_solver3(prod, nx=96, ny=64, km=32, a_imp=a_imp, use_logp=use_logp, last_call=last_call, fp_out=fp_out)


def test_nh_update_dz_d(prod):
    N.check_update_dz_d(prod)
    N.check_update_dz_d(prod, lev_over=dict(nord=2, do_vort_damp=True, vtdm4=0.06), hord=8)
    N.check_update_dz_d(prod, nx=33, ny=9, km=3)
    N.check_update_dz_d(prod, nx=96, ny=96, km=16)


@pytest.mark.parametrize("hydrostatic", [False, True])
def test_nh_p_grad_c(prod, hydrostatic):
    N.check_p_grad_c(prod, hydrostatic=hydrostatic)


def test_nh_p_grad(prod):
    N.check_nh_p_grad(prod)
    N.check_nh_p_grad(prod, nx=33, ny=9, km=3)
    N.check_nh_p_grad(prod, nx=96, ny=96, km=16)


def test_one_grad_p_in_the_nonhydrostatic_loop(prod, tmp_path):
    """beta < -0.1 (dyn_core.F90:939, :1029-1030, :1909-2030 with hydrostatic = .false.): Riem_Solver3 leaves the full pressure, one_grad_p
    with a2b_ord4 of delp as the layer weights takes the place of nh_p_grad -- the kernel, then the substep loops, doubly periodic
    (with and without the external-mode damping) and on the sphere"""
    N.check_one_grad_p_nh(prod)
    N.check_one_grad_p_nh(prod, nx=96, ny=96, km=16, d_ext=0.0)
    D.check_substeps(prod, n_split=3, flags=dict(beta=-1.0))
    D.check_substeps(prod, n_split=2, flags=dict(beta=-1.0, d_ext=0.0, a_imp=0.75))
    cs, gs = PC.CC.sphere(25)
    for t in (1, 5):
        N.check_one_grad_p_nh(prod, km=4, grid=gs[t], d_ext=0.0)
    assert max(PC.check_substeps_nh(prod, npx=25, npz=5, n_split=3, flags=dict(beta=-1.0)).values()) <= 1e-13
    # ... and through the Fortran host (fv3_host_mod's loop), bit-identical to the Python host
    import fortran_host as F
    assert "fv3_solo: done" in F.check_fortran_host(prod, tmp_path, nx=24, ny=16, npz=8, nq=1, beta=-1.0)
    assert "fv3_solo_sphere: done" in F.check_fortran_sphere(prod, tmp_path, npx=13, npz=8, nq=1, n_split=2, k_split=1, beta=-1.0)


def test_split_p_grad_and_grad1_p_update(prod):
    """beta > 0 (dyn_core.F90:1795-1900, :2033-2116): kernels over three calls, both substep loops, doubly periodic and sphere"""
    N.check_split_p_grad(prod)
    N.check_split_p_grad(prod, nx=96, ny=96, km=16, beta=0.25)
    N.check_grad1_p_update(prod)
    N.check_grad1_p_update(prod, nx=96, ny=96, km=16, d_ext=0.0)
    D.check_substeps(prod, n_split=3, flags=dict(beta=0.4, a_imp=0.6))
    D.check_substeps_hydrostatic(prod, n_split=3, flags=dict(beta=0.4))
    D.check_fv_step(prod, flags=dict(beta=0.3))
    cs, gs = PC.CC.sphere(25)
    for t in (0, 4):
        N.check_split_p_grad(prod, km=4, grid=gs[t])
        N.check_grad1_p_update(prod, km=4, grid=gs[t], d_ext=0.0)
    assert max(PC.check_substeps_nh(prod, npx=25, npz=5, n_split=3, flags=dict(beta=0.4)).values()) <= 1e-13
    assert max(PC.check_substeps_hydrostatic(prod, npx=25, npz=4, n_split=3, flags=dict(beta=0.4)).values()) <= 1e-13
    r = PC.check_jw_step(prod, npx=25, npz=20, k_split=2, n_split=2, bdt=900.0, hydrostatic=False, flags=dict(beta=0.4))
    assert r.pop("finite") == 1.0 and max(r.values()) <= 1e-12


def test_inline_q(prod):
    """inline_q (sw_core.F90:1020-1043): the tracers inside d_sw every substep, nonhydrostatic and hydrostatic, with and without the
    del-n damping whose mass is d_sw's half-updated delp; doubly periodic (marching and tile kernels) and sphere (hybrid frame)"""
    D.check_fv_step(prod, n_split=3, flags=dict(inline_q=True))
    D.check_fv_step(prod, nx=96, ny=64, nq=3, flags=dict(inline_q=True, do_vort_damp=True, vtdm4=0.06, nord=2, hord_tr=5))
    D.check_fv_step_hydrostatic(prod, nx=96, ny=64, n_split=3, flags=dict(inline_q=True))
    D.check_fv_step_hydrostatic(prod, flags=dict(inline_q=True, do_vort_damp=True, vtdm4=0.06, nord=1))
    r = PC.check_jw_step(prod, npx=49, npz=12, k_split=2, n_split=2, bdt=600.0, hydrostatic=False, nq=2, flags=dict(inline_q=True))
    assert r.pop("finite") == 1.0 and max(r.values()) <= 1e-12
    r = PC.check_jw_step(prod, npx=25, npz=12, k_split=1, n_split=2, bdt=900.0, hydrostatic=True, nq=2,
                         flags=dict(inline_q=True, do_vort_damp=True, vtdm4=0.06, nord=2))
    assert r.pop("finite") == 1.0 and max(r.values()) <= 1e-12


def test_fill2d(prod):
    """fill2D (fv_fill.F90:183-258): the kernels against the oracle, then after tracer_2d in whole steps (hord_tr < 8, moist_phys)"""
    assert T.check_fill2d(prod) <= 1e-15
    assert T.check_fill2d(prod, nx=192, ny=96, npz=16) <= 1e-15
    D.check_fv_step(prod, flags=dict(hord_tr=5), fill2d=(0, 1), q_range=(-0.3, 1.0))
    r = PC.check_jw_step(prod, npx=25, npz=12, k_split=2, n_split=2, bdt=900.0, hydrostatic=False, nq=2, flags=dict(hord_tr=5),
                         fill2d=(0,), q_shift=1.0)
    assert r.pop("finite") == 1.0 and max(r.values()) <= 1e-12


def test_prt_maxmin_and_the_reference_timers(prod):
    """f4: prt_mxm (tools/fv_diagnostics.F90:4265-4313) -- max, min and g_sum's area mean of the last level (fv_grid_utils.F90:2879-2925),
    restated here in the reference's loop order -- and fv3_profile's events under the reference's timing_on / timing_off names"""
    import math
    from gfdl_atmos_cubed_sphere_amd.dyn_core import DynCore, DynFlags
    from gfdl_atmos_cubed_sphere_amd.layout import Bounds
    from gfdl_atmos_cubed_sphere_amd.lib import Context
    bd = Bounds(1, 37, 1, 22)
    g = P.make_grid(bd, True)
    rng = np.random.default_rng(8)
    q = np.asfortranarray(rng.normal(250.0, 30.0, bd.shape("A", 7)))
    area = np.asarray(g.m["area"])
    ctx = Context(g, 7, lib=prod)
    try:
        got = ctx.prt_maxmin(ctx.from_host(q), 0.01)
        c = (slice(bd.ng, bd.ng + bd.nx), slice(bd.ng, bd.ng + bd.ny))
        gsum = 0.0
        for j in range(bd.ng, bd.ng + bd.ny):          # g_sum's "quick local sum": do j; do i
            for i in range(bd.ng, bd.ng + bd.nx):
                gsum = gsum + q[i, j, 6] * area[i, j]
        garea = math.fsum(area[c].ravel())
        assert got[0] == q[c].max() * 0.01 and got[1] == q[c].min() * 0.01
        assert abs(got[2] - gsum / garea * 0.01) <= 4e-16 * abs(got[2])
        # the timers: a substep loop under fv3_profile, reported under the reference's names
        st, dp0 = D.make_state(Bounds(1, 37, 1, 22), 7)
        dc = DynCore(ctx, DynFlags(n_split=2, ptop=N.PTOP), dp0)
        dc.set_state(st["u"], st["v"], st["w"], st["delp"], st["pt"], st["delz"], st["phis"])
        ctx.profile(True)
        dc.run(4.0)
        per_kernel = ctx.profile_report()
        dc.run(4.0)
        timers = ctx.profile_report_timers()
        ctx.profile(False)
        assert set(timers) <= {"C_SW", "D_SW", "UPDATE_DZ_C", "UPDATE_DZ", "Riem_Solver", "PG_D", "COMM_TOTAL", "tracer_2d", "Fill2D", "Remapping",
                               "DYN_CORE"}
        assert {"C_SW", "D_SW", "UPDATE_DZ_C", "UPDATE_DZ", "Riem_Solver", "PG_D", "COMM_TOTAL"} <= set(timers)
        assert sum(n for n, _ in timers.values()) == sum(n for n, _ in per_kernel.values())     # every launch under exactly one timer
        assert timers["Riem_Solver"][0] == per_kernel["riem_solver3"][0] + per_kernel["riem_solver_c"][0] == 4
    finally:
        ctx.close()


def test_remap_in_lds_and_in_slabs(prod):
    """Lagrangian_to_Eulerian with the column in LDS (csrc/remap_fast.h: levels across the lanes, the spline's elimination in the
    reference's order by hand-over rounds, the limiters' curvature re-formed from a one-byte code) -- the default where it is built, and
    BIT-IDENTICAL to the oracle like the slab kernels (csrc/remap_kernels.h), which the same cases run through with
    FV3_MI355X_REMAP_LDS=0 and which keep what the LDS kernels are not built for (kord_tm > 0, remap_te)"""
    for kw in (dict(), dict(km=20, nx=33, ny=9), dict(hydrostatic=True), dict(last_step=True, adiabatic=False),
               dict(hydrostatic=True, last_step=True, adiabatic=False, kord_tm=-10, kord=10), dict(kord=9, kord_tm=-9, nq=7),
               dict(km=127, nx=17, ny=3, nq=1), dict(km=79, nq=4, kord=13, kord_tm=-14), dict(kord=15, kord_tm=-15, km=8)):
        assert R.check_remap(prod, **kw) <= 1e-14
        assert R.check_remap(prod, lds=False, **kw) <= 1e-14
    # round 4, second half: use_cond / moist_kappa in the LDS kernels too (RemapFastScalars<false, true>: cappa from moist_cv of the
    # un-remapped tracers in the temperature transform, of the remapped ones in pkz, the last step's conversion with the condensates)
    for kw in (dict(moist_kappa=True), dict(moist_kappa=True, use_cond=True, last_step=True, kord_tm=-8, nwat=6, adiabatic=False),
               dict(moist_kappa=False, use_cond=True, last_step=True, kord_tm=-8, nwat=6, adiabatic=False),
               dict(moist_kappa=True, use_cond=True, kord_tm=-9, nwat=3, km=79, nx=17, ny=3)):
        assert R.check_remap(prod, **kw) == 0.0
        assert R.check_remap(prod, lds=False, **kw) == 0.0
    # ... and flagstruct%fill: fillz on the column in LDS, one thread per column that holds a negative value
    for kw in (dict(fill=True), dict(fill=True, nq=7, kord=9, kord_tm=-9), dict(fill=True, km=79, nx=17, ny=3, nq=3),
               dict(fill=True, moist_kappa=True, use_cond=True, nwat=6)):
        assert R.check_remap(prod, **kw) == 0.0
        assert R.check_remap(prod, lds=False, **kw) == 0.0
    # 5 levels per lane up to km = 79, 8 from km = 80 (RemapFastCoreT<L>): both sides of the switch, many tracers on few levels
    for kw in (dict(km=79, nx=17, ny=3, nq=3), dict(km=78, nx=17, ny=3, nq=1, hydrostatic=True), dict(km=80, nx=17, ny=3, nq=1),
               dict(km=5, nx=17, ny=3), dict(km=79, nq=9, kord=9, kord_tm=-9, last_step=True, adiabatic=False)):
        assert R.check_remap(prod, **kw) == 0.0
    assert R.check_remap(prod, kord_tm=9) <= 1e-14


def test_remap_te(prod):
    """flagstruct%remap_te (fv_mapz.F90:232-286, :348-360, :576-619, :655-663): total energy through map_scalar (kord_tm /= 0) or
    map1_cubic (kord_tm = 0), T_v and pkz from it; columns, then whole steps on both domains"""
    for kw in (dict(), dict(kord_tm=9), dict(kord_tm=0), dict(hydrostatic=True), dict(hydrostatic=True, kord_tm=0),
               dict(last_step=True, adiabatic=False), dict(hydrostatic=True, last_step=True, adiabatic=False, kord_tm=10),
               dict(moist_kappa=True), dict(moist_kappa=True, use_cond=True, last_step=True, adiabatic=False)):
        assert R.check_remap(prod, nx=70, ny=33, km=20, remap_te=True, **kw) <= 1e-14
    D.check_fv_step(prod, remap_te=True)
    D.check_fv_step(prod, remap_te=True, kord_tm=0, nq=0)
    D.check_fv_step_hydrostatic(prod, remap_te=True)
    D.check_fv_cycle_consv(prod, remap_te=True)                      # te_2d of the energy fixer from the remapped energy (:655-663)
    D.check_fv_cycle_consv(prod, hydrostatic=True, remap_te=True)
    r = PC.check_jw_step(prod, npx=25, npz=12, k_split=2, n_split=2, bdt=900.0, hydrostatic=False, nq=1, remap_te=True)
    assert r.pop("finite") == 1.0 and max(r.values()) <= 1e-12
    r = PC.check_jw_step(prod, npx=25, npz=12, k_split=1, n_split=2, bdt=900.0, hydrostatic=True, remap_te=True, kord_tm=0)
    assert r.pop("finite") == 1.0 and max(r.values()) <= 1e-12


def test_nh_halos_and_geopk(prod):
    N.check_halos_and_geopk(prod)


# ---- whole acoustic substeps ---------------------------------------------------------------------
import parity_dyn as D


def test_dyn_core_substeps(prod):
    D.check_substeps(prod, n_split=2)


def test_dyn_core_substeps_sim_solver_damping(prod):
    D.check_substeps(prod, n_split=3, flags=dict(a_imp=0.75, nord=2, do_vort_damp=True, vtdm4=0.06, dddmp=0.2))


def test_dyn_core_substeps_larger(prod):
    D.check_substeps(prod, nx=96, ny=64, npz=32, n_split=2)


# ---- vertical remap -----------------------------------------------------------------------------
import parity_remap as R


@pytest.mark.parametrize("hydrostatic,last_step,kord_tm,kord,nq", [(False, False, -8, 8, 2), (False, True, -9, 9, 6),
                                                                    (True, False, -8, 8, 1), (False, False, 8, 10, 0),
                                                                    (True, True, 10, 11, 3), (False, True, -10, 13, 2),
                                                                    (False, True, -14, 14, 3), (True, False, 15, 15, 7),
                                                                    (False, False, -15, 14, 6), (False, True, 12, 12, 4),
                                                                    (True, False, 12, 12, 7), (False, True, -11, 11, 3),
                                                                    (True, False, -11, 11, 2), (False, False, -12, 12, 3),
                                                                    (True, True, -12, 12, 2)])
def test_remap(prod, hydrostatic, last_step, kord_tm, kord, nq):
    # |kord| = 11, 12 with kord_tm < 0 (tie-sensitive limiters on the transformed temperature) are in the matrix since the
    # kernels and the oracle share one exp / log (include/fv3_math.h)
    R.check_remap(prod, hydrostatic=hydrostatic, last_step=last_step, kord_tm=kord_tm, kord=kord, nq=nq)


@pytest.mark.parametrize("hydrostatic,last_step,kord_tm,kord,nq", [(False, False, -7, 7, 3), (True, True, 7, 7, 6), (False, True, -6, 6, 2),
                                                                    (True, False, 5, 5, 7), (False, False, -4, 4, 3), (False, True, 4, 3, 2),
                                                                    (True, False, -9, -8, 2), (False, True, -10, 6, 0)])
def test_remap_ppm_profile(prod, hydrostatic, last_step, kord_tm, kord, nq):
    """kord <= 7: the map routines take ppm_profile + ppm_limiters (fv_operators.F90:1382-1723) instead of cs_profile: Huynh's
    2nd constraint (7), the positive-definite / full-monotonicity / standard limiters (6, 5, 4, 3), for winds, w, T (|kord_tm|)
    and tracers (iv = -1, -2, 1, 0); a NEGATIVE kord_mt / kord_tr also fails the reference's "kord > 7" test and lands there"""
    R.check_remap(prod, hydrostatic=hydrostatic, last_step=last_step, kord_tm=kord_tm, kord=kord, nq=nq)


@pytest.mark.parametrize("nt,nq,kord", [(1, 4, 9), (2, 5, 10), (3, 4, 9), (3, 5, 8), (3, 7, 10), (3, 7, 11), (3, 3, 13)])
def test_remap_tracer_groups(prod, nt, nq, kord, monkeypatch):
    """tracers remapped side by side in groups of up to nt per thread (remap_tracers_col): even dealing (4 = 2 + 2,
    7 = 3 + 2 + 2), both tracer forms (nq <= 5, nq > 5), |kord| = 11 falling back to one tracer at a time"""
    monkeypatch.setenv("FV3_MI355X_REMAP_NT", str(nt))
    R.check_remap(prod, nq=nq, kord=kord, last_step=True)
    R.check_remap(prod, nq=nq, kord=kord, hydrostatic=True, kord_tm=-kord if kord != 13 else -10, fill=(nq == 7))


def test_remap_larger(prod):
    R.check_remap(prod, nx=96, ny=64, km=32, nq=2)


# ---- tracer_2d -------------------------------------------------------------------------------------
import parity_tracer as T


def test_tracer_2d(prod):
    T.check_tracer_2d(prod)
    _, nsplt = T.check_tracer_2d(prod, big_courant=True, hord=-5)
    assert nsplt > 1
    T.check_tracer_2d(prod, q_split=2, trdm=0.06, nord_tr=1, hord=10)
    T.check_tracer_2d(prod, nx=33, ny=9, npz=7, nq=7, big_courant=True)
    T.check_tracer_2d(prod, nx=96, ny=96, npz=16, nq=4)


@pytest.mark.gpu
@pytest.mark.parametrize("nt", [1, 2, 3, 4])
def test_tracer_2d_tracers_per_wavefront(prod, nt, monkeypatch):
    """the sub-cycle kernel with 1..4 tracers per wavefront (short last group, finished levels, sub-cycling)"""
    monkeypatch.setenv("FV3_MI355X_TRACER_NT", str(nt))
    T.check_tracer_2d(prod, nx=70, ny=21, npz=3, nq=5, big_courant=True)
    T.check_tracer_2d(prod, nx=96, ny=96, npz=16, nq=7, hord=10)


def test_torch_alias_of_device_array_and_exchange_path(prod):
    """The multi-GPU halo path aliases library-owned buffers as torch tensors (__cuda_array_interface__) and
    packs/unpacks with strided torch ops: check the alias is zero-copy, Fortran-strided and coherent with the
    kernels, by running the exchange code with a single rank (local periodic copies) against the device kernel."""
    import torch
    from gfdl_atmos_cubed_sphere_amd.grid import doubly_periodic
    from gfdl_atmos_cubed_sphere_amd.halo import HaloTopology, exchange_tensors
    from gfdl_atmos_cubed_sphere_amd.layout import Bounds
    bd = Bounds(1, 20, 1, 12)
    g = doubly_periodic(bd, 21, 13)
    ctx = L.Context(g, 3, lib=prod, stream=torch.cuda.current_stream().cuda_stream)
    try:
        rng = np.random.default_rng(2)
        for kind in ("A", "U", "V", "B"):
            a = np.asfortranarray(rng.uniform(-1, 1, bd.shape(kind, 3)))
            d1, d2 = ctx.from_host(a), ctx.from_host(a)
            t = torch.as_tensor(d1, device="cuda")
            assert t.data_ptr() == d1.ptr and tuple(t.shape) == d1.shape and t.stride()[0] == 1
            exchange_tensors(HaloTopology(bd, 1, 1, 0), [(t, kind)])
            torch.cuda.synchronize()
            ctx.halo_fill_periodic(d2, kind)
            assert np.array_equal(d1.download(), d2.download()), kind
    finally:
        ctx.close()


@pytest.mark.parametrize("nq,k_split", [(2, 2), (0, 1)])
def test_fv_dynamics_step(prod, nq, k_split):
    D.check_fv_step(prod, nq=nq, k_split=k_split)


@pytest.mark.parametrize("hord,hord_mt", [(10, 10), (8, 6), (5, 5), (6, 8)])
def test_d_sw_multi_strip_march(prod, hord, hord_mt):
    """several 58-column strips and several row segments of the wave-marching kernels"""
    P.check_d_sw(prod, nx=130, ny=100, npz=3, par_over=dict(hord_dp=hord, hord_tm=hord, hord_vt=hord, hord_mt=hord_mt))


@pytest.mark.parametrize("hydrostatic", [False, True])
@pytest.mark.parametrize("hord,hord_mt", [(10, 10), (8, 6), (5, 5), (-5, 8), (6, 8)])
def test_d_sw_uniform_metrics(prod, hord, hord_mt, hydrostatic):
    """Cartesian doubly periodic gridstruct (Grid::geom == 2): the kernels that carry the metric terms as scalars"""
    P.check_d_sw(prod, nx=130, ny=64, npz=3, perturb=False, hydrostatic=hydrostatic,
                 par_over=dict(hord_dp=hord, hord_tm=hord, hord_vt=hord, hord_mt=hord_mt))


@pytest.mark.parametrize("perturb", [False, "ortho"])
def test_geometry_modes_off_same_result(prod, perturb, monkeypatch):
    """FV3_MI355X_GEOM=0 sends an orthogonal / uniform gridstruct through the general kernels: same parity"""
    monkeypatch.setenv("FV3_MI355X_GEOM", "0")
    P.check_c_sw(prod, nx=70, ny=30, npz=2, perturb=perturb)
    P.check_d_sw(prod, nx=70, ny=30, npz=2, perturb=perturb)


@pytest.mark.parametrize("nx,ny,hydro", [(130, 100, False), (55, 44, True)])
def test_c_sw_multi_strip_march(prod, nx, ny, hydro):
    P.check_c_sw(prod, nx=nx, ny=ny, npz=2, hydrostatic=hydro)


def test_update_dz_d_and_tracers_multi_strip_march(prod):
    import parity_tracer as T
    N.check_update_dz_d(prod, nx=130, ny=100, km=3)
    T.check_tracer_2d(prod, nx=130, ny=100, npz=3, nq=2)
    T.check_tracer_2d(prod, nx=70, ny=60, npz=3, nq=2, big_courant=True)


def test_halo_pack_unpack_kernels(prod):
    P.check_halo_packed(prod)


def test_c384_tile_vs_oracle(prod):
    """the C384 horizontal tile (7 strips x 8 segments of the marching kernels), a few levels: bit-level parity"""
    assert P.check_c_sw(prod, nx=384, ny=384, npz=5) <= P.TOL
    assert max(P.check_d_sw(prod, nx=384, ny=384, npz=5).values()) <= P.TOL


@pytest.mark.parametrize("nx", [384, 1024])
def test_c384l127_flux_form_properties(prod, nx):
    """BASELINE sizes 384 x 384 x 127 (configs[2]'s tile) and the WHOLE 1024 x 1024 x 127 doubly periodic domain of configs[3] on one
    GPU, on the GPU alone: size-independent properties of the pair -- mass conservation of d_sw (sw_core.F90:1059-1060) and consistency
    of the flux capacitors with the delp update"""
    import numpy as np
    from fields import smooth_state
    from gfdl_atmos_cubed_sphere_amd.halo import HaloExchanger
    from gfdl_atmos_cubed_sphere_amd.layout import Bounds
    from gfdl_atmos_cubed_sphere_amd.lib import Context
    from test_oracle_properties import default_levels
    npz = 127
    bd = Bounds(1, nx, 1, nx)
    g = P.make_grid(bd, False)   # constant metrics: a truly periodic tile
    ctx = Context(g, npz, lib=prod)
    try:
        halo = HaloExchanger(ctx, 1, 1, 0, 1)
        st = smooth_state(bd, npz, noise=0.05)
        d = {k: ctx.from_host(v) for k, v in st.items()}
        for n, kind in P.CSW_OUT:
            d[n] = ctx.zeros(kind, npz)
        for n, kind in (("mfx", "FX"), ("mfy", "FY"), ("cx", "CX"), ("cy", "CY"), ("crx", "CX"), ("cry", "CY"),
                        ("xfx", "CX"), ("yfx", "CY"), ("delp_out", "A"), ("pt_out", "A"), ("u_out", "U"),
                        ("v_out", "V"), ("w_out", "A"), ("heat_s", "CC"), ("diss_e", "CC")):
            d[n] = ctx.zeros(kind, npz)
        ctx.dsw_levels(default_levels(npz))
        par = dict(P.DSW_PAR)
        par.update(hydrostatic=0, use_cond=0)
        dt = par["dt"]
        halo.update([(d["delp"], "A"), (d["pt"], "A"), (d["w"], "A"), (d["u"], "U"), (d["v"], "V")])  # periodic state
        st["delp"] = d["delp"].download()
        ctx.c_sw(d["delpc"], d["delp"], d["ptc"], d["pt"], d["u"], d["v"], d["w"], d["uc"], d["vc"], d["ua"], d["va"],
                 d["wc"], d["ut"], d["vt"], d["divg_d"], 1, 0.5 * dt, False)
        halo.update([(d["uc"], "V"), (d["vc"], "U"), (d["divg_d"], "B")])
        ctx.d_sw(par, None, d["delp"], d["pt"], d["u"], d["v"], d["w"], d["uc"], d["vc"], d["ua"], d["va"], d["divg_d"],
                 d["mfx"], d["mfy"], d["cx"], d["cy"], d["crx"], d["cry"], d["xfx"], d["yfx"], None, d["delp_out"],
                 d["pt_out"], d["u_out"], d["v_out"], d["w_out"], None, d["heat_s"], d["diss_e"])
        r = (bd.is_, bd.ie, bd.js, bd.je)
        dp0 = bd.view(st["delp"], "A", *r)
        dp1 = bd.view(d["delp_out"].download(), "A", *r)
        area = bd.view(g.area, "A", *r)[:, :, None]
        m0, m1 = np.sum(dp0 * area, axis=(0, 1)), np.sum(dp1 * area, axis=(0, 1))
        assert np.all(np.isfinite(dp1)) and np.max(np.abs(m1 - m0) / m0) < 1e-13        # per level, periodic domain
        mfx, mfy = d["mfx"].download(), d["mfy"].download()                                # started from zero
        div = (mfx[:-1] - mfx[1:] + mfy[:, :-1] - mfy[:, 1:]) * bd.view(g.rarea, "A", *r)[:, :, None]
        assert np.max(np.abs((dp1 - dp0) - div)) <= 1e-12 * np.max(np.abs(dp0))
        for n in ("pt_out", "w_out", "u_out", "v_out"):
            assert np.all(np.isfinite(d[n].download())), n
    finally:
        ctx.close()


@pytest.mark.parametrize("hydro,n_con,nmax", [(False, None, 2), (True, None, 3), (False, 2, 1)])
def test_heat_source_path(prod, hydro, n_con, nmax):
    N.check_heat_source_path(prod, hydrostatic=hydro, n_con=n_con, nmax=nmax)


def test_dyn_core_substeps_with_dissipative_heating(prod):
    D.check_substeps(prod, n_split=2, flags=dict(d_con=1.0, do_vort_damp=True, vtdm4=0.06, nord=2))
    D.check_substeps(prod, n_split=2, flags=dict(d_con=0.5))


@pytest.mark.parametrize("d_ext", [0.02, 0.0])
def test_one_grad_p_hydrostatic(prod, d_ext):
    N.check_one_grad_p(prod, d_ext=d_ext)


def test_dyn_core_substeps_hydrostatic(prod):
    D.check_substeps_hydrostatic(prod)
    D.check_substeps_hydrostatic(prod, nx=70, ny=60, npz=6, flags=dict(d_ext=0.0))
    D.check_substeps_hydrostatic(prod, n_split=3, flags=dict(d_con=1.0, do_vort_damp=True, vtdm4=0.06, nord=2))


def test_fv_dynamics_step_hydrostatic(prod):
    D.check_fv_step_hydrostatic(prod)


def test_fv_dynamics_cycle_from_temperature(prod):
    D.check_fv_cycle_from_temperature(prod)


def test_baseline_config1_shape_hydrostatic(prod):
    """BASELINE configs[0] shape (doubly periodic 48 x 48 x 32, hydrostatic): two substeps vs the oracle"""
    D.check_substeps_hydrostatic(prod, nx=48, ny=48, npz=32, n_split=2)


def test_d_sw_interior_then_rest_equals_d_sw(prod):
    """3 x 3 strips/segments: the interior box first, the frame afterwards (halo-exchange overlap form)"""
    assert max(P.check_d_sw(prod, nx=130, ny=100, npz=3, phases=True).values()) <= P.TOL
    assert max(P.check_d_sw(prod, nx=130, ny=100, npz=3, hydrostatic=True, phases=True).values()) <= P.TOL
    assert max(P.check_d_sw(prod, nx=40, ny=19, npz=3, phases=True).values()) <= P.TOL      # no interior: rest does all


@pytest.mark.parametrize("nx,ny", [(6, 5), (58, 48), (59, 49), (117, 97), (8, 64), (61, 4)])
def test_march_strip_and_segment_boundaries(prod, nx, ny):
    """tiny tiles, exactly one strip / segment, one column / row more than a strip / segment, ragged last ones"""
    assert P.check_c_sw(prod, nx=nx, ny=ny, npz=2) <= P.TOL
    assert max(P.check_d_sw(prod, nx=nx, ny=ny, npz=3).values()) <= P.TOL
    assert max(P.check_d_sw(prod, nx=nx, ny=ny, npz=3, hydrostatic=True, phases=True).values()) <= P.TOL


@pytest.mark.parametrize("state", ["rest", "tophat", "checker"])
@pytest.mark.parametrize("hord", [10, 5])
def test_limiter_branch_point_states(prod, state, hord):
    """no wind + constant scalars (all Courant numbers / slopes exactly zero), top-hats, 2-cell oscillations"""
    assert P.check_c_sw(prod, nx=64, ny=40, npz=2, state=state) <= P.TOL
    over = dict(hord_dp=hord, hord_tm=hord, hord_vt=hord, hord_mt=hord)
    assert max(P.check_d_sw(prod, nx=64, ny=40, npz=3, state=state, par_over=over).values()) <= P.TOL


def test_geometry_mode_detection(prod):
    """fv3_grid_upload classifies the gridstruct from its arrays (general / orthogonal / orthogonal + uniform)."""
    from gfdl_atmos_cubed_sphere_amd.lib import Context
    from gfdl_atmos_cubed_sphere_amd.layout import Bounds
    bd = Bounds(1, 12, 1, 9)
    for perturb, want in ((True, 0), ("ortho", 1), (False, 2)):
        ctx = Context(P.make_grid(bd, perturb), 2, lib=prod)
        try:
            assert ctx.geom == want
        finally:
            ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("a_imp", [1.0, 0.75])
def test_riem_solvers_fast_tau_w_sec(prod, a_imp):
    """fast_tau_w_sec > 0: the Rayleigh damping of w inside SIM1_solver / SIM_solver (nh_utils.F90:356-367, :1363-1371, :1498-1506),
    the levels-across-the-lanes kernels and the slab kernels, dry and moist"""
    N.check_riem_solver3(prod, a_imp=a_imp, tau_w=25.0)
    N.check_riem_solver3(prod, a_imp=a_imp, tau_w=25.0, lds=False, use_logp=True, last_call=True, fp_out=True)
    N.check_riem_solver3(prod, a_imp=a_imp, tau_w=40.0, use_cond=True, moist_kappa=True, nx=40, ny=9, km=19)
    if a_imp > 0.999:
        N.check_riem_solver_c(prod, tau_w=25.0)
        N.check_riem_solver_c(prod, tau_w=25.0, lds=False)
        N.check_riem_solver_c(prod, tau_w=25.0, use_cond=True, nx=40, ny=9, km=19)


@pytest.mark.gpu
@pytest.mark.parametrize("dims", [dict(), dict(nx=33, ny=9, km=3), dict(nx=70, ny=35, km=37), dict(nx=384, ny=96, km=127), dict(nx=64, ny=32, km=16), dict(nx=31, ny=15, km=17)])
def test_nh_p_grad_in_one_kernel(prod, dims):
    """NhPGradFused (a2b_ord4 of pp, pk, gz, delp and nh_p_grad in one kernel: the corner values never leave LDS) against the oracle and,
    bit for bit, against the two-kernel path; tiles and layer chunks that end inside the domain / the column"""
    N.check_nh_p_grad_fused_bits(prod, **dims)


def test_consv_am(prod):
    """flagstruct%consv_am: compute_aam before and after the k_split loop, the reproducible sums, u00 and the wind correction
    (fv_dynamics.F90:358-361, :747-800, :1266-1314)"""
    N.check_consv_am_kernels(prod)
    D.check_fv_cycle_from_temperature(prod, consv_am=True)


def test_substeps_with_fast_tau_w_sec_and_rf_fast(prod):
    """the acoustic substeps with the Rayleigh damping of w inside the solvers and Ray_fast at their end (dyn_core.F90:536, :940, :1057-1060)"""
    npz = 10
    pfull = N.PTOP * 1.2 + (1.0e5 - N.PTOP) * (np.arange(npz) + 0.5) / npz
    fl = dict(fast_tau_w_sec=40.0, rf_fast=True, tau=0.002, rf_cutoff=float(pfull[4]) + 1.0)
    D.check_substeps(prod, npz=npz, n_split=3, bdt=6.0, flags=fl, pfull=pfull, ks=7)
    D.check_substeps(prod, npz=npz, n_split=2, bdt=4.0, flags=dict(fl, a_imp=0.75, rf_fast=False), pfull=pfull, ks=7)


@pytest.mark.parametrize("hydrostatic", [False, True])
def test_ray_fast(prod, hydrostatic):
    """Ray_fast (RF_fast, dyn_core.F90:1057-1060, :2485-2601), bit for bit; k_rf beyond kmax and no level above the cutoff too"""
    kmax, k_rf = N.check_ray_fast(prod, hydrostatic=hydrostatic)
    assert kmax == 7 and k_rf == 7
    N.check_ray_fast(prod, hydrostatic=hydrostatic, nx=64, ny=5, km=7, ks=2)            # k_rf < kmax
    assert N.check_ray_fast(prod, hydrostatic=hydrostatic, rf_cutoff=10.0) == (1, 0)    # nothing above the cutoff: rf = 1, nothing moves ...


@pytest.mark.parametrize("hydrostatic,conserve", [(False, True), (True, True), (False, False)])
def test_c2l_and_rayleigh_friction(prod, hydrostatic, conserve):
    """fv_dynamics around the k_split loop: cubed_to_latlon (ord 2, 4) and Rayleigh_Friction, grid_type = 4"""
    N.check_c2l_and_rayleigh(prod, hydrostatic=hydrostatic, conserve=conserve)


def test_fv_dynamics_call_with_rayleigh_friction(prod):
    """T -> pkz, Rayleigh_Friction, theta_v, k_split loop, last remap back to T, cubed_to_latlon"""
    D.check_fv_cycle_from_temperature(prod, tau=0.01)


@pytest.mark.parametrize("hydrostatic", [False, True])
def test_mix_dp(prod, hydrostatic):
    """mix_dp (flagstruct%fill_dp, dyn_core.F90:820, :2119-2200) against the oracle, bit for bit; then inside the substep loop"""
    N.check_mix_dp(prod, hydrostatic=hydrostatic)
    if hydrostatic:
        D.check_substeps_hydrostatic(prod, flags=dict(fill_dp=True))
    else:
        D.check_substeps(prod, flags=dict(fill_dp=True), akbk="thin")


def test_registry_forget(prod):
    """a host array that is freed and allocated again at the same address leaves the lazy registry"""
    P.check_registry_forget(prod)


def test_dyn_core_substeps_with_do_diss_est(prod):
    """flagstruct%do_diss_est through DynCore: d_sw's diss_e of every level summed into diss_est over the substeps (dyn_core.F90:805-811);
    with the heating on top (d_con = 1) in the second run"""
    D.check_substeps(prod, do_diss_est=True)
    D.check_substeps(prod, do_diss_est=True, flags=dict(d_con=1.0), npz=10)
    D.check_substeps_hydrostatic(prod, do_diss_est=True)


def test_fv_dynamics_call_with_rf_fast(prod):
    """flagstruct%tau > 0 with RF_fast given to FvDynamics: no Rayleigh_Friction (fv_dynamics.F90:362), Ray_fast after every acoustic
    substep instead (dyn_core.F90:1057-1060) -- ONE tau for both (with tau in two places this ran undamped)"""
    D.check_fv_cycle_from_temperature(prod, tau=0.002, rf_fast=True)


def test_halo_messages_through_rccl_loopback():
    """the N-GPU message path (pack -> RCCL batch_isend_irecv -> unpack) on one GPU: every message is a self message"""
    import subprocess
    import sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "rccl_loopback_worker.py"), "29541"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "rccl loopback ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_fortran_host_drives_the_library(prod, tmp_path):
    """Host orchestration in Fortran (fv3_host_mod: dyn_core, tracer_2d, the k_split loop) over ISO_C_BINDING -> C ABI ->
    HIP, compiled with the image's amdflang and run on the GPU: the state equals the Python host's bit for bit"""
    import fortran_host as F
    if F.fortran_compiler() is None:
        pytest.skip("no Fortran compiler in this image")
    out = F.check_fortran_host(prod, tmp_path)
    assert "fv3_solo: done" in out
    # ... and with every group halo update as RCCL self messages posted by the library (fv3_halo_start / _complete)
    out = F.check_fortran_host(prod, tmp_path, host_comm=True)
    assert "fv3_solo: done" in out
    # the hydrostatic branch and the dissipative heating of both branches
    assert "fv3_solo: done" in F.check_fortran_host(prod, tmp_path, npz=12, nq=1, hydrostatic=True, d_con=1.0)
    assert "fv3_solo: done" in F.check_fortran_host(prod, tmp_path, npz=12, nq=0, hydrostatic=False, d_con=1.0)
    assert "fv3_solo: done" in F.check_fortran_host(prod, tmp_path, npz=8, nq=2, hydrostatic=False, inline_q=True)
    assert "fv3_solo: done" in F.check_fortran_host(prod, tmp_path, npz=8, nq=1, hydrostatic=True, inline_q=True, beta=0.3)
    # thermostruct%use_cond / moist_kappa through the Fortran loop: q_con in d_sw and the Riemann solvers, cappa, the moist remap
    assert "fv3_solo: done" in F.check_fortran_host(prod, tmp_path, npz=8, nq=7, use_cond=True, moist_kappa=True, d_con=1.0)
    assert "fv3_solo: done" in F.check_fortran_host(prod, tmp_path, npz=8, nq=6, use_cond=True)


def test_fortran_host_on_the_cubed_sphere(prod, tmp_path):
    """the Fortran host on grid_type = 0 (fv3_sphere_mod + fv3_solo_sphere): six contexts in one process, every cube-edge message of
    dyn_core / tracer_2d as RCCL send / recv behind the C ABI (fv3_cube_halo_start / _complete), mpp_get_boundary, adv_pe: a C24
    Jablonowski-Williamson fv_dynamics call bit-identical to the Python host's (device gathers) on every face"""
    import fortran_host as F
    if F.fortran_compiler() is None:
        pytest.skip("no Fortran compiler in this image")
    assert "fv3_solo_sphere: done" in F.check_fortran_sphere(prod, tmp_path, npx=25, npz=20, nq=2, hydrostatic=False)
    assert "fv3_solo_sphere: done" in F.check_fortran_sphere(prod, tmp_path, npx=25, npz=20, nq=0, hydrostatic=True, d_con=1.0, k_split=1)
    assert "fv3_solo_sphere: done" in F.check_fortran_sphere(prod, tmp_path, npx=25, npz=8, nq=0, hydrostatic=False, beta=0.4, n_split=3)
    assert "fv3_solo_sphere: done" in F.check_fortran_sphere(prod, tmp_path, npx=25, npz=8, nq=2, hydrostatic=False, inline_q=True)
    assert "fv3_solo_sphere: done" in F.check_fortran_sphere(prod, tmp_path, npx=25, npz=8, nq=1, hydrostatic=False, remap_te=True)


def test_fortran_dyn_core_with_the_reference_argument_list(prod, tmp_path):
    """fv3_dyn_core_mod's dyn_core: the reference's argument list (model/dyn_core.F90:94-98: host arrays with the fv_arrays bounds,
    gridstruct / flagstruct / bd by their reference names) over the device-resident loop; both branches, with the heating"""
    import fortran_host as F
    if F.fortran_compiler() is None:
        pytest.skip("no Fortran compiler in this image")
    assert "fv3_solo_refsig: done" in F.check_fortran_refsig(prod, tmp_path)
    assert "fv3_solo_refsig: done" in F.check_fortran_refsig(prod, tmp_path, npz=12, d_con=1.0)
    assert "fv3_solo_refsig: done" in F.check_fortran_refsig(prod, tmp_path, npz=12, hydrostatic=True, d_con=1.0)
    assert "fv3_solo_refsig: done" in F.check_fortran_refsig(prod, tmp_path, npz=8, beta=0.4)
    assert "fv3_solo_refsig: done" in F.check_fortran_refsig(prod, tmp_path, npz=8, nsteps=3, registry=True)    # the lazy host-address registry
    assert "fv3_solo_refsig: done" in F.check_fortran_refsig(prod, tmp_path, npz=8, hydrostatic=True, d_con=1.0, registry=True)
    assert "fv3_solo_refsig: done" in F.check_fortran_refsig(prod, tmp_path, npz=10, fast_tau_w_sec=40.0, rf_fast_tau=0.002)   # fast_tau_w_sec, RF_fast
    assert "fv3_solo_refsig: done" in F.check_fortran_refsig(prod, tmp_path, npz=10, hydrostatic=True, rf_fast_tau=0.002)
    assert "fv3_solo_refsig: done" in F.check_fortran_refsig(prod, tmp_path, npz=8, hydrostatic=True, beta=0.4)
    assert "fv3_solo_refsig: done" in F.check_fortran_refsig(prod, tmp_path, npz=8, moist=True, d_con=1.0)     # thermostruct%use_cond / moist_kappa
    assert "fv3_solo_refsig: done" in F.check_fortran_refsig(prod, tmp_path, npz=8, do_diss_est=True, d_con=1.0)   # flagstruct%do_diss_est: diss_est in and out
    assert "fv3_solo_refsig: done" in F.check_fortran_refsig(prod, tmp_path, npz=8, hydrostatic=True, do_diss_est=True)
    assert "fv3_solo_refsig: done" in F.check_fortran_refsig(prod, tmp_path, npz=8, fill_dp=True)          # flagstruct%fill_dp: mix_dp after d_sw
    assert "fv3_solo_refsig: done" in F.check_fortran_refsig(prod, tmp_path, npz=8, hydrostatic=True, fill_dp=True)
    # fv_dynamics with ITS reference argument list (model/fv_dynamics.F90:79-85): T -> theta_v, the k_split loop with tracers and
    # the remap, last_step, cubed_to_latlon
    assert "fv3_solo_refsig: done" in F.check_fortran_fv_dynamics(prod, tmp_path)
    assert "fv3_solo_refsig: done" in F.check_fortran_fv_dynamics(prod, tmp_path, hydrostatic=True, npz=12, nq=1)
    # thermostruct%use_cond = moist_kappa = .true. -- the reference's DEFAULTS (fv_arrays.F90:1226-1227) -- through the reference-signature
    # fv_dynamics: the water species by get_tracer_index, q_con / cappa formed by moist_cv inside, q_con handed back
    assert "fv3_solo_refsig: done" in F.check_fortran_fv_dynamics(prod, tmp_path, nq=7, moist=True)
    assert "fv3_solo_refsig: done" in F.check_fortran_fv_dynamics(prod, tmp_path, nq=6, moist=True, consv_te=1.0, npz=10)
    assert "fv3_solo_refsig: done" in F.check_fortran_fv_dynamics(prod, tmp_path, nq=1, do_diss_est=True)     # diss_est out of fv_dynamics
    assert "fv3_solo_refsig: done" in F.check_fortran_fv_dynamics(prod, tmp_path, nq=0, consv_am=True)        # flagstruct%consv_am
    assert "fv3_solo_refsig: done" in F.check_fortran_fv_dynamics(prod, tmp_path, nq=1, beta=-1.0, hybrid_z=True)             # one_grad_p in the nonhydrostatic loop (beta < -0.1); hybrid_z = .true. is accepted (the reference never reads it)


@pytest.mark.parametrize("use_cond,moist_kappa", [(True, False), (True, True), (False, True)])
@pytest.mark.parametrize("a_imp", [1.0, 0.75])
def test_riem_solvers_moist(prod, use_cond, moist_kappa, a_imp):
    """use_cond / moist_kappa branches of Riem_Solver3 (nh_core.F90:96-166) and Riem_Solver_c (nh_utils.F90:383-438)"""
    N.check_riem_solver3(prod, a_imp=a_imp, use_cond=use_cond, moist_kappa=moist_kappa)
    if a_imp > 0.999 and use_cond:
        N.check_riem_solver_c(prod, use_cond=True, moist_kappa=moist_kappa)


@pytest.mark.parametrize("moist_kappa,use_cond,last_step,kord_tm,nwat", [(True, True, False, -9, 6), (True, True, True, -8, 6),
                                                                          (False, True, True, -8, 6), (True, False, False, 9, 3)])
def test_remap_moist(prod, moist_kappa, use_cond, last_step, kord_tm, nwat):
    """moist_kappa / use_cond branches of Lagrangian_to_Eulerian (fv_mapz.F90:212-219, :463-478, :806-811) with moist_cv"""
    R.check_remap(prod, moist_kappa=moist_kappa, use_cond=use_cond, last_step=last_step, kord_tm=kord_tm, nwat=nwat,
                  adiabatic=False)


@pytest.mark.parametrize("kw", [dict(), dict(with_qv=False), dict(hydrostatic=True), dict(split=True),
                                dict(moist_kappa=True, use_cond=True), dict(moist_kappa=True, use_cond=True, split=True),
                                dict(use_cond=True), dict(hydrostatic=True, use_cond=True)])
def test_pt_to_theta_v(prod, kw):
    """T -> theta_v before the k_split loop (fv_dynamics.F90:296-329, :379-399), dry / zvir / moist_kappa / use_cond"""
    N.check_pt_to_theta_v(prod, **kw)


@pytest.mark.parametrize("moist_kappa", [True, False])
def test_fv_dynamics_call_moist(prod, moist_kappa):
    """whole fv_dynamics call with use_cond (+ moist_kappa): moist_cv conversions, q_con through d_sw and the Riemann
    solvers, moist remap, T on return"""
    D.check_fv_cycle_moist(prod, moist_kappa=moist_kappa)


def test_fv_dynamics_call_moist_heating(prod):
    """the dissipative heating of dyn_core with moist_kappa: pkz from the per-cell cappa (dyn_core.F90:1338-1340)"""
    D.check_fv_cycle_moist(prod, moist_kappa=True, flags=dict(d_con=1.0, do_vort_damp=True, vtdm4=0.06, nord=2))


@pytest.mark.parametrize("nq", [2, 6])
def test_remap_fillz(prod, nq):
    """flagstruct%fill: fillz (fv_fill.F90:34-137) on the remapped tracers, both tracer remap forms (nq <= 5, nq > 5)"""
    R.check_remap(prod, nq=nq, fill=True)


def test_baseline_config2_tile_shape(prod):
    """one 96 x 96 x 79 tile (the face size of BASELINE configs[1], C96L79, hydrostatic) on a Cartesian and on a general
    gridstruct: c_sw and d_sw against the oracle"""
    for perturb in (False, True):
        P.check_c_sw(prod, nx=96, ny=96, npz=79, hydrostatic=True, perturb=perturb)
        P.check_d_sw(prod, nx=96, ny=96, npz=79, hydrostatic=True, perturb=perturb)


def test_baseline_config1_test_case_1(prod):
    """BASELINE configs[0]: doubly periodic 48 x 48 x 32, hydrostatic, the reference's test_case = 1 initial condition
    (uniform flow carrying a block of mass): one dt_atmos of the k_split loop (substeps, tracer_2d, remap) vs the oracle"""
    D.check_fv_step_hydrostatic(prod, nx=48, ny=48, npz=32, nq=1, k_split=1, n_split=3, bdt=6.0, ic="test_case_1")


@pytest.mark.parametrize("hord", [7, 9, 11, 12, 13])
def test_tracer_2d_positive_definite_schemes(prod, hord):
    """hord_tr = 9 / 13 (pert_ppm), 11 (ppm_fac slopes), 12 (Lin & Rood positive definite), tp_core.F90:604-641: the marching
    kernels (one and three tracers per wavefront) and, with the first sub-cycle damped, the tile kernel"""
    T.check_tracer_2d(prod, nq=4, hord=hord, big_courant=True)
    T.check_tracer_2d(prod, nx=70, ny=21, npz=3, nq=1, hord=hord)
    T.check_tracer_2d(prod, q_split=2, trdm=0.06, nord_tr=1, hord=hord)


def test_d_sw_interior_does_not_read_halos_in_flight(prod, monkeypatch):
    """uc, vc halos poisoned during 'interior' and restored before 'rest' (the exchange is in flight then): ragged last
    strips / segments of 1-3 cells / rows must keep the interior launch off the halo"""
    for nx, ny in ((130, 100), (117, 100), (175, 100)):
        assert max(P.check_d_sw(prod, nx=nx, ny=ny, npz=3, phases="poison").values()) <= P.TOL
    monkeypatch.setenv("FV3_MI355X_MARCH_TJ_FUSED", "8")
    for nx, ny in ((130, 97), (131, 26), (118, 98), (119, 99)):
        assert max(P.check_d_sw(prod, nx=nx, ny=ny, npz=3, phases="poison").values()) <= P.TOL


# ---- the column path at BASELINE depth (L79 = configs 2 and 5, L127 = configs 3 and 4): per-wavefront blocked scratch
# ---- slabs, register budgets and unrolling of the k loops are exercised at the depth the configurations run at
@pytest.mark.parametrize("km", [79, 127])
def test_nh_columns_at_baseline_depth(prod, km):
    dims = dict(nx=200, ny=72, km=km)          # 225 wavefronts of columns, ragged last one
    N.check_update_dz_c(prod, **dims)
    N.check_riem_solver_c(prod, **dims)
    N.check_riem_solver_c(prod, a_imp=0.75, **dims)
    N.check_riem_solver3(prod, **dims)
    N.check_riem_solver3(prod, a_imp=0.75, use_logp=True, last_call=True, fp_out=True, **dims)
    N.check_update_dz_d(prod, **dims)
    N.check_halos_and_geopk(prod, **dims)
    N.check_nh_p_grad(prod, **dims)
    N.check_p_grad_c(prod, **dims)


@pytest.mark.parametrize("km,nq,hydrostatic,last_step,kord", [(127, 4, False, True, 9), (127, 4, False, False, 10),
                                                              (79, 33, True, False, 9), (79, 33, True, True, 8),
                                                              (79, 6, True, True, 11)])
def test_remap_at_baseline_depth(prod, km, nq, hydrostatic, last_step, kord):
    """Lagrangian_to_Eulerian at L127 with 4 tracers (config 3) and at L79 with 33 (config 5)"""
    R.check_remap(prod, nx=200, ny=72, km=km, nq=nq, hydrostatic=hydrostatic, last_step=last_step, kord=kord,
                  kord_tm=-kord if kord != 10 else 10)


def test_config3_c384l127_whole_nh_step(prod):
    """BASELINE configs[2] size: one 384 x 384 x 127 tile, nonhydrostatic (Riem_Solver3 / SIM1 path), one k_split cycle of
    fv_dynamics (2 acoustic substeps, tracer_2d with 2 tracers, Lagrangian_to_Eulerian) against the oracle"""
    D.check_fv_step(prod, nx=384, ny=384, npz=127, nq=2, k_split=1, n_split=2, bdt=8.0)


def test_config4_1024_wide_strip_nh_substeps(prod):
    """BASELINE configs[3]: a 1024-wide strip of the 1024 x 1024 x 127 doubly periodic nonhydrostatic domain (18 strips of
    wavefronts in x), two acoustic substeps against the oracle"""
    D.check_substeps(prod, nx=1024, ny=64, npz=127, n_split=2)


def test_config5_c768l79_hydrostatic_step_33_tracers(prod):
    """BASELINE configs[4] size in x and z on a reduced j extent: 768 x 96 x 79, hydrostatic, 33 tracers: one k_split cycle
    (substeps, tracer_2d, remap) against the oracle"""
    D.check_fv_step_hydrostatic(prod, nx=768, ny=96, npz=79, nq=33, k_split=1, n_split=2, bdt=8.0)


@pytest.mark.parametrize("state", ["westward", "swirl"])
@pytest.mark.parametrize("hydrostatic", [False, True])
def test_reversed_and_mixed_winds_three_strips(prod, state, hydrostatic):
    """u < 0 / winds of both signs on 3 strips x 2-3 segments: every upwind select takes the other neighbour, also in the
    first and last lanes a strip owns (the default states have u > 0 everywhere)"""
    for perturb in (False, True):
        assert P.check_c_sw(prod, nx=130, ny=70, npz=3, hydrostatic=hydrostatic, perturb=perturb, state=state) <= P.TOL
        assert max(P.check_d_sw(prod, nx=130, ny=70, npz=3, hydrostatic=hydrostatic, perturb=perturb, state=state).values()) <= P.TOL


def test_reversed_winds_whole_substeps_and_tracers(prod):
    D.check_substeps(prod, nx=130, ny=30, npz=6, n_split=2, bdt=8.0, ic="westward")
    D.check_substeps_hydrostatic(prod, nx=96, ny=24, npz=6, n_split=2, bdt=8.0)
    D.check_substeps_hydrostatic(prod, nx=130, ny=30, npz=6, n_split=2, bdt=8.0, ic="westward")
    T.check_tracer_2d(prod, nx=130, ny=30, npz=3, nq=3, reverse=True)
    T.check_tracer_2d(prod, nx=130, ny=30, npz=3, nq=4, reverse=True, big_courant=True)


# ---- cubed sphere (grid_type < 3): the pass kernels against the oracle, all six faces ----------------------------------
import parity_cubed as PC


@pytest.mark.parametrize("hydrostatic", [False, True])
def test_cubed_c_sw(prod, hydrostatic):
    assert PC.check_c_sw(prod, npx=13, npz=3, hydrostatic=hydrostatic) <= P.TOL
    assert PC.check_c_sw(prod, npx=49, npz=4, hydrostatic=hydrostatic) <= P.TOL          # C48 faces


@pytest.mark.parametrize("hord", [10, 8, 5, -5, 6, 7, 9, 11, 12, 13])
def test_cubed_fv_tp_2d(prod, hord):
    assert PC.check_fv_tp_2d(prod, hord, npx=25) <= P.TOL
    assert PC.check_fv_tp_2d(prod, hord, npx=25, mass_flux=True, faces=(2, 5)) <= P.TOL


@pytest.mark.parametrize("kw", [dict(hydrostatic=True), dict(hydrostatic=True, flags=dict(nord=2)),
                                dict(hydrostatic=True, flags=dict(nord=3)),
                                dict(hydrostatic=True, par_over=dict(hord_mt=5, hord_vt=5, hord_tm=5, hord_dp=5)),
                                dict(hydrostatic=True, par_over=dict(hord_mt=6, hord_vt=6, hord_tm=6, hord_dp=-5)),
                                dict(hydrostatic=True, par_over=dict(hord_mt=8, hord_vt=8, hord_tm=8, hord_dp=8)),
                                dict(hydrostatic=True, par_over=dict(hord_mt=9)),
                                dict(hydrostatic=False, flags=dict(n_sponge=-1))])
def test_cubed_d_sw(prod, kw):
    assert max(PC.check_d_sw(prod, npx=25, npz=3, **kw).values()) <= P.TOL


def test_cubed_c48_pair(prod):
    """a gnomonic C48 face set with all edges and corners: c_sw and d_sw bit for bit against the oracle"""
    assert max(PC.check_d_sw(prod, npx=49, npz=4, hydrostatic=True).values()) <= P.TOL


def test_cubed_d_sw_nonhydrostatic_default(prod):
    assert max(PC.check_d_sw(prod, npx=25, npz=4, hydrostatic=False).values()) <= P.TOL


def test_cubed_sphere_nonhydrostatic_substeps(prod):
    """the nonhydrostatic substep loop on the whole sphere (six contexts on one GPU) against the six-face oracle"""
    assert max(PC.check_substeps_nh(prod, npx=25, npz=6, n_split=2).values()) <= 1e-12
    assert max(PC.check_substeps_nh(prod, npx=49, npz=8, n_split=3, bdt=450.0).values()) <= 1e-12


def test_cubed_a2b_ord4_through_the_pressure_gradients(prod):
    cs, gs = PC.CC.sphere(25)
    for t in (0, 3, 5):
        N.check_nh_p_grad(prod, km=8, grid=gs[t])
        N.check_one_grad_p(prod, km=8, grid=gs[t], d_ext=0.0)


def test_cubed_sphere_hydrostatic_substeps(prod):
    """the hydrostatic acoustic substep loop on the whole sphere on one GPU (C24 and C48, six contexts, device halo gathers)"""
    assert max(PC.check_substeps_hydrostatic(prod, npx=25, npz=6, n_split=2).values()) <= 1e-13
    assert max(PC.check_substeps_hydrostatic(prod, npx=49, npz=8, n_split=3, bdt=450.0).values()) <= 1e-13


def test_baseline_config2_c96l79_jablonowski_williamson(prod):
    """BASELINE configs[1]: C96L79 Jablonowski-Williamson baroclinic wave (test_case = 13), hydrostatic, the whole cubed
    sphere on one MI355X (six contexts, device halo gathers): one dt_atmos = k_split 2 x (n_split 3 acoustic substeps +
    Lagrangian_to_Eulerian) against the six-face orchestration of the oracle, rel-RMS < 1e-12 on every prognostic field"""
    r = PC.check_jw_step(prod, npx=97, npz=79, k_split=2, n_split=3, bdt=900.0)
    assert r.pop("finite") == 1.0 and max(r.values()) <= 1e-12


def test_cubed_sphere_jw_c24(prod):
    r = PC.check_jw_step(prod, npx=25, npz=79, k_split=1, n_split=2, bdt=900.0)
    assert r.pop("finite") == 1.0 and max(r.values()) <= 1e-12


def test_config3_small_c96_l127_nonhydrostatic_jw(prod):
    """BASELINE configs[2] at the size the oracle reaches: the C96 L127 nonhydrostatic baroclinic wave on the whole sphere (six
    contexts on one MI355X), one remap cycle of three acoustic substeps, < 1e-12 against the six-face oracle"""
    r = PC.check_jw_step(prod, npx=97, npz=127, k_split=1, n_split=3, bdt=450.0, hydrostatic=False)
    assert r.pop("finite") == 1.0 and max(r.values()) <= 1e-12


def test_config3_c384_l127_nonhydrostatic_sphere(prod):
    """BASELINE configs[2] at full size: C384 L127 nonhydrostatic on one MI355X (6 x 384 x 384 x 127 cells).  Past the oracle's
    reach: finite, global air mass kept to rounding, the winds of the shared cube edges equal on both faces, the state moved"""
    r = PC.check_sphere_properties(prod, npx=385, npz=127, hydrostatic=False, k_split=1, n_split=2, bdt=75.0)
    assert r["finite"] == 1.0 and r["mass_drift"] < 1e-13 and r["edge_mismatch"] == 0.0 and r["moved"] > 1e-3, r


def test_config2_c96_l79_sphere_properties(prod):
    r = PC.check_sphere_properties(prod, npx=97, npz=79, hydrostatic=True, k_split=2, n_split=6, bdt=1800.0)
    assert r["finite"] == 1.0 and r["mass_drift"] < 1e-13 and r["edge_mismatch"] == 0.0 and r["moved"] > 1e-3, r


@pytest.mark.parametrize("kw", [dict(npx=25, nq=3), dict(npx=25, courant_scale=40.0, hord=5, nq=2), dict(npx=49, npz=6, q_split=2, hord=13, nq=2)])
def test_cubed_tracer_2d(prod, kw):
    assert PC.check_tracer_2d(prod, **kw)["q"] <= P.TOL


def test_config5_small_c96_l79_sphere_with_33_tracers(prod):
    """BASELINE configs[4] at the size the oracle reaches, six faces on one GPU: C96 L79 + 33 advected tracers (fv_tracer2d),
    one remap cycle, < 1e-12 against the six-face oracle"""
    r = PC.check_jw_step(prod, npx=97, npz=79, k_split=1, n_split=3, bdt=450.0, hydrostatic=True, nq=33)
    assert r.pop("finite") == 1.0 and max(r.values()) <= 1e-12


def test_config5_c192_l79_sphere_with_33_tracers_properties(prod):
    """past the oracle's reach: air mass and the mass of every tracer kept to rounding on the whole sphere"""
    r = PC.check_sphere_properties(prod, npx=193, npz=79, hydrostatic=True, k_split=1, n_split=3, bdt=225.0, nq=33)
    assert r["finite"] == 1.0 and r["tracer_finite"] == 1.0 and r["mass_drift"] < 1e-13 and r["tracer_mass_drift"] < 1e-12, r


def test_cubed_sphere_faces_on_their_own_streams(prod):
    """six contexts on six HIP streams (their kernels overlap on the GPU), joined and forked around every halo gather: the same
    numbers as on one stream -- nonhydrostatic JW step with tracers on C48 L20 against the six-face oracle"""
    r = PC.check_jw_step(prod, npx=49, npz=20, k_split=2, n_split=2, bdt=900.0, hydrostatic=False, nq=3, face_streams=True)
    assert r.pop("finite") == 1.0 and max(r.values()) <= 1e-12


def test_exchange_behind_the_c_abi_rccl_self_messages(prod):
    """fv3_comm_init (ncclCommInitRank through the run-time loaded librccl) + fv3_halo_start / fv3_halo_complete: on one GPU
    every one of the 8 messages of a group is an RCCL send / recv to this same rank on the context's communication stream;
    halos equal the periodic fill, a substep loop driven through it equals the oracle"""
    from gfdl_atmos_cubed_sphere_amd.halo import HaloExchanger
    from gfdl_atmos_cubed_sphere_amd.layout import Bounds, periodic_fill
    from gfdl_atmos_cubed_sphere_amd.lib import Context
    bd = Bounds(1, 40, 1, 24)
    g = P.make_grid(bd, False)
    ctx = Context(g, 5, lib=prod)
    try:
        hx = HaloExchanger(ctx, 1, 1, 0, 1, native=True)
        rng = np.random.default_rng(2)
        host = {k: np.asfortranarray(rng.uniform(-1, 1, bd.shape(k, 5))) for k in ("A", "U", "V", "B")}
        dev = {k: ctx.from_host(v) for k, v in host.items()}
        for _ in range(3):
            hx.update([(dev[k], k) for k in ("A", "U", "V", "B")])
        for k, v in host.items():
            ref = v.copy(order="F")
            for n in range(5):
                periodic_fill(bd, ref[:, :, n], k)
            assert np.array_equal(dev[k].download(), ref), k
        assert np.array_equal(ctx.allreduce_max(np.array([3.0, -1.0])), [3.0, -1.0])
    finally:
        ctx.close()


def test_substeps_through_the_c_abi_exchange(prod):
    from gfdl_atmos_cubed_sphere_amd.dyn_core import DynCore, DynFlags
    from gfdl_atmos_cubed_sphere_amd.halo import HaloExchanger
    from gfdl_atmos_cubed_sphere_amd.layout import Bounds
    from gfdl_atmos_cubed_sphere_amd.lib import Context
    import oracle_dyn_core as OD
    nx, ny, npz = 40, 24, 8
    bd = Bounds(1, nx, 1, ny)
    g = P.make_grid(bd, False)
    st, dp0 = D.make_state(bd, npz)
    fl = DynFlags(n_split=2, ptop=N.PTOP)
    ref = OD.run(g, npz, fl, dp0, st, 4.0)
    ctx = Context(g, npz, lib=prod)
    try:
        dc = DynCore(ctx, fl, dp0, halo=HaloExchanger(ctx, 1, 1, 0, 1, native=True, split_single=True))
        dc.set_state(st["u"], st["v"], st["w"], st["delp"], st["pt"], st["delz"], st["phis"])
        dc.run(4.0)
        got = dc.get_state()
        for n, kind, rr in (("u", "U", (bd.is_, bd.ie, bd.js, bd.je + 1)), ("v", "V", (bd.is_, bd.ie + 1, bd.js, bd.je)),
                            ("w", "A", (bd.is_, bd.ie, bd.js, bd.je)), ("delp", "A", (bd.is_, bd.ie, bd.js, bd.je)),
                            ("pt", "A", (bd.is_, bd.ie, bd.js, bd.je))):
            P.assert_close(n, bd.view(got[n], kind, *rr), bd.view(ref[n], kind, *rr), 1e-13)
    finally:
        ctx.close()


# ---- cubed sphere: the damping / heating branches a production namelist switches on ------------------------------------------------
PROD = dict(do_vort_damp=True, vtdm4=0.06, nord=3, d_con=1.0, dddmp=0.5)


@pytest.mark.parametrize("hydrostatic", [False, True])
def test_cubed_c384_pair_vs_oracle(prod, hydrostatic):
    """the strip / segment / frame ownership arithmetic of the cubed hybrid (CswMarch<2,0,true> + FramePass + Tp2dFrameFused
    + the marching d_sw kernels with their store masks) is size-dependent, and C384 (7 strips x 6-8 segments, the width BASELINE
    config 3 runs at) was held to the oracle only through mass / edge properties.  Six gnomonic C384 faces, c_sw -> halo -> d_sw:
    two sponge + two regular levels with the default flags (marching interior + frame passes), and 18 levels with the production
    damping set (every level damped: the del-2n chains, the flux-form march and the frame kernel), sw_core.F90:79-1606."""
    assert PC.check_c_sw(prod, npx=385, npz=4, hydrostatic=hydrostatic) <= P.TOL
    assert max(PC.check_d_sw(prod, npx=385, npz=4, hydrostatic=hydrostatic).values()) <= P.TOL
    assert max(PC.check_d_sw(prod, npx=385, npz=18, hydrostatic=hydrostatic, faces=(0, 3, 5), flags=PROD,
                             par_over=dict(dddmp=0.5)).values()) <= P.TOL


@pytest.mark.parametrize("lane_d2", ["1", "0"])
def test_cubed_pair_in_two_lanes(prod, monkeypatch, lane_d2):
    """round 4: the frame / sponge-level passes of a cubed-sphere face on the side stream BESIDE the marching kernels (fv3_api.hip
    dsw_cubed / csw_cubed).  Forced on (FV3_MI355X_SIDE_STREAM=2: by default only a face that launches on its own and is large takes
    the two lanes) and held to the oracle on six C96 / C384 faces -- a race between the lanes would show as a wrong frame --, then the
    six faces as a group through whole nonhydrostatic substeps."""
    monkeypatch.setenv("FV3_MI355X_SIDE_STREAM", "2")
    monkeypatch.setenv("FV3_MI355X_LANE_D2", lane_d2)
    for rep in range(3):
        assert PC.check_c_sw(prod, npx=97, npz=16, hydrostatic=False) <= P.TOL
        assert max(PC.check_d_sw(prod, npx=97, npz=16, hydrostatic=False).values()) <= P.TOL
    assert PC.check_c_sw(prod, npx=385, npz=4, hydrostatic=False) <= P.TOL
    assert max(PC.check_d_sw(prod, npx=385, npz=6, hydrostatic=False).values()) <= P.TOL
    assert max(PC.check_d_sw(prod, npx=385, npz=6, hydrostatic=True, faces=(0, 3)).values()) <= P.TOL
    assert max(PC.check_substeps_nh(prod, npx=49, npz=12, n_split=2).values()) <= 1e-12
    # every level damped (production namelist): the momentum half up to the absolute vorticity beside the transport half
    for rep in range(2):
        assert max(PC.check_d_sw(prod, npx=97, npz=18, hydrostatic=False, faces=(0, 3, 5), flags=PROD, par_over=dict(dddmp=0.5)).values()) <= P.TOL
    assert max(PC.check_d_sw(prod, npx=385, npz=18, hydrostatic=True, faces=(1,), flags=PROD, par_over=dict(dddmp=0.5)).values()) <= P.TOL


@pytest.mark.parametrize("hydrostatic", [True, False])
def test_cubed_d_sw_damping_fused_chains(prod, hydrostatic):
    """C32 / C48 faces: the del-2n chains as one LDS-tile launch away from the corners (cubed_damp.h DelnFused), the damped whole-face
    levels through the fused transport with delp's damping fluxes as an input (cubed_tpf.h); equal to the oracle like the passes"""
    assert max(PC.check_d_sw(prod, npx=33, npz=17, hydrostatic=hydrostatic, faces=(0, 5), flags=PROD, par_over=dict(dddmp=0.5)).values()) <= P.TOL
    assert max(PC.check_d_sw(prod, npx=49, npz=5, hydrostatic=hydrostatic, faces=(2,),
                             flags=dict(do_vort_damp=True, vtdm4=0.06, nord=1)).values()) <= P.TOL


@pytest.mark.parametrize("hydrostatic", [True, False])
@pytest.mark.parametrize("kw", [dict(flags=dict(do_vort_damp=True, vtdm4=0.06, nord=2)), dict(flags=dict(d_con=1.0)),
                                dict(flags=dict(dddmp=0.2, nord=2), par_over=dict(dddmp=0.2)),
                                dict(flags=PROD, par_over=dict(dddmp=0.5))])
def test_cubed_d_sw_damping_and_heating(prod, kw, hydrostatic):
    assert max(PC.check_d_sw(prod, npx=25, npz=12, hydrostatic=hydrostatic, **kw).values()) <= P.TOL


@pytest.mark.parametrize("kw", [dict(flags=dict(d_con=1.0), grid_flags=dict(do_diss_est=True, prevent_diss_cooling=False)),
                                dict(flags=dict(), grid_flags=dict(do_diss_est=True, prevent_diss_cooling=True)),
                                dict(flags=PROD, par_over=dict(dddmp=0.5), grid_flags=dict(do_diss_est=True, prevent_diss_cooling=True))])
def test_cubed_d_sw_dissipation_estimate(prod, kw):
    """do_diss_est on a cubed-sphere face (sw_core.F90:964-978, :1462, :1516-1586): diss_est of every level, with and without the
    heating, the vorticity damping and the cooling limiter; the work arrays are zeroed where nothing damps the vorticity"""
    assert max(PC.check_d_sw(prod, npx=25, npz=6, hydrostatic=False, faces=(1, 4), **kw).values()) <= P.TOL


@pytest.mark.parametrize("hydrostatic,conserve", [(False, True), (True, True), (False, False)])
def test_cubed_sphere_rayleigh_friction(prod, hydrostatic, conserve):
    """Rayleigh_Friction on the six faces: u2f through the cubed-sphere cubed_to_latlon, its halo across the cube edges, heating +
    implicit damping of u, v, w (fv_dynamics.F90:1126-1264)"""
    assert PC.check_rayleigh(prod, npx=25, hydrostatic=hydrostatic, conserve=conserve) <= 1e-14


@pytest.mark.parametrize("moist_kappa", [True, False])
def test_cubed_sphere_moist_fv_dynamics_call(prod, moist_kappa):
    """use_cond (+ moist_kappa) through a whole nonhydrostatic fv_dynamics call on the six faces: moist_cv conversion, q_con in d_sw
    and both Riemann solvers with its halo across the cube edges, moist remap, back to T (SURVEY 8(f) item 3 on the sphere)"""
    r = PC.check_jw_step_moist(prod, npx=25, npz=20, moist_kappa=moist_kappa)
    assert max(r.values()) <= 1e-12


@pytest.mark.parametrize("hydrostatic,ideal", [(False, False), (True, False), (False, True)])
def test_cubed_sphere_rayleigh_super(prod, hydrostatic, ideal):
    """Rayleigh_Super, the form fv_dynamics applies on the cubed sphere for tau > 0 (fv_dynamics.F90:362-366, :953-1124), through the
    host's dispatch on the six faces; is_ideal_case: relaxation towards the winds of the first call"""
    assert PC.check_rayleigh_super(prod, npx=25, hydrostatic=hydrostatic, ideal=ideal) <= 1e-14


@pytest.mark.parametrize("kw", [dict(), dict(hydrostatic=True), dict(hydrostatic=True, adiabatic=True), dict(adiabatic=True),
                                dict(consv_te=-2.0), dict(consv_te=-2.0, hydrostatic=True)])
def test_total_energy_conservation(prod, kw):
    """consv_te: compute_total_energy before the loop, the energy fixer of the last remap (te_2d, zsum0 / zsum1, the reproducing
    global sums, dtmp) and the final T_v -> T step with dtmp; a prescribed flux for consv_te < 0; the energy of the final state
    closes on the initial one"""
    assert max(D.check_fv_cycle_consv(prod, **kw).values()) <= 1e-12


def test_ordered_sum_is_exact_and_order_independent(prod):
    """fv3_ordered_sum (g_sum with reproduce = .true.: the extended-fixed-point sum) against math.fsum and the host's own
    implementation (global_sum.py), on addends spread over 23 orders of magnitude, permuted and split"""
    import math
    from gfdl_atmos_cubed_sphere_amd.global_sum import reproducing_sum
    from gfdl_atmos_cubed_sphere_amd.layout import Bounds
    from gfdl_atmos_cubed_sphere_amd.lib import Context
    ctx = Context(P.make_grid(Bounds(1, 8, 1, 8), False), 5, lib=prod)
    try:
        rng = np.random.default_rng(1)
        a = rng.normal(0, 1e9, 200000) * rng.choice([1e-14, 1.0, 1e9], 200000)
        s = ctx.ordered_sum(a)
        assert abs(s - math.fsum(a)) <= abs(s) * 2.3e-16
        p_ = rng.permutation(a)
        assert ctx.ordered_sum(p_) == s == reproducing_sum([p_[:777], p_[777:90000], p_[90000:]])
        # the range of the extended fixed point format: both implementations refuse what FMS aborts on (|a| >= 2**138), and a
        # leading digit that would not survive the int64 all-reduce
        for bad in (np.array([1.0, 2.0 ** 138]), np.full(3000, 2.0 ** 131)):
            with pytest.raises(Exception):
                ctx.ordered_sum(bad)
            with pytest.raises(OverflowError):
                reproducing_sum([bad])
        assert ctx.ordered_sum(np.array([2.0 ** 130, 1.0, -2.0 ** 130])) == 1.0 == reproducing_sum([np.array([2.0 ** 130, 1.0, -2.0 ** 130])])
    finally:
        ctx.close()


def test_cubed_sphere_total_energy_conservation(prod):
    """the same on the six faces (hydrostatic JW): global sums over the sphere"""
    assert max(PC.check_jw_consv(prod, npx=25).values()) <= 1e-12


def test_cubed_adv_pe(prod):
    """the advective term of the omega diagnostic on the six faces (adv_pe, dyn_core.F90:1195, :1529-1632)"""
    assert PC.check_adv_pe(prod, npx=25) <= 1e-14


def test_config4_supercell_initial_condition(prod):
    """BASELINE configs[3]'s initial condition (doubly periodic supercell, test_case = 17: Weisman-Klemp sounding, sheared wind, warm
    bubble, vapour) through a whole nonhydrostatic fv_dynamics call against the oracle loop; air mass to rounding, the bubble rises"""
    assert max(D.check_supercell_step(prod, nx=96, ny=64, npz=64).values()) <= 1e-12


def test_cubed_del2_cubed_and_damped_transports(prod):
    assert PC.check_del2_cubed(prod, npx=25, npz=4, nmax=3) <= P.TOL
    for kw in (dict(nord=2, damp_c=0.05), dict(nord=2, damp_c=0.05, mass_flux=True)):
        assert PC.check_fv_tp_2d(prod, 8, npx=25, faces=(0, 2, 5), **kw) <= P.TOL
    assert PC.check_tracer_2d(prod, npx=25, nord_tr=2, trdm=0.1, courant_scale=40.0, hord=5, nq=2)["q"] <= P.TOL


def test_cubed_sphere_with_production_flags(prod):
    """nord = 3, do_vort_damp, d_con = 1, dddmp = 0.5 on the whole sphere: substeps on C24 / C48 faces (the hybrid keeps the damped
    levels on the pass kernels), a nonhydrostatic JW step with tracers on C48 L79, and the conservation properties on C96 L79"""
    assert max(PC.check_substeps_hydrostatic(prod, npx=25, npz=12, n_split=2, flags=PROD).values()) <= 1e-13
    assert max(PC.check_substeps_nh(prod, npx=49, npz=12, n_split=2, bdt=450.0, flags=PROD).values()) <= 1e-12
    r = PC.check_jw_step(prod, npx=49, npz=79, k_split=1, n_split=3, bdt=450.0, hydrostatic=False, nq=2, flags=PROD)
    assert r.pop("finite") == 1.0 and max(r.values()) <= 1e-12
    r = PC.check_sphere_properties(prod, npx=97, npz=79, hydrostatic=False, k_split=1, n_split=3, bdt=450.0, nq=2, flags=PROD)
    assert r["finite"] == 1.0 and r["mass_drift"] < 1e-13 and r["tracer_mass_drift"] < 1e-12 and r["edge_mismatch"] == 0.0, r


def test_cubed_sphere_hydrostatic_external_mode_damping(prod):
    assert max(PC.check_substeps_hydrostatic(prod, npx=49, npz=12, n_split=2, bdt=450.0, flags=dict(d_ext=0.02)).values()) <= 1e-13


@pytest.mark.parametrize("hydrostatic", [True, False])
def test_cubed_d_sw_use_cond(prod, hydrostatic):
    """thermostruct%use_cond on a face: q_con transported with delp's mass fluxes (and pt's damping when it is on)"""
    assert max(PC.check_d_sw(prod, npx=25, npz=12, hydrostatic=hydrostatic, faces=(0, 4), use_cond=True).values()) <= P.TOL
    assert max(PC.check_d_sw(prod, npx=25, npz=12, hydrostatic=hydrostatic, faces=(2,), use_cond=True,
                             flags=dict(do_vort_damp=True, vtdm4=0.06, nord=2)).values()) <= P.TOL


@pytest.mark.parametrize("c2l_ord", [2, 4])
def test_cubed_to_latlon_on_the_sphere(prod, c2l_ord):
    """cubed_to_latlon on the six faces (the a11 .. a22 rotation of init_cubed_to_latlon, the two-point forms next to the face
    edges): device = oracle, and the result is the analytic (east, north) wind of the test state to discretisation error"""
    assert PC.check_c2l(prod, c2l_ord, npx=25) <= P.TOL


def test_cube_table_of_the_library_equals_the_oracle(prod):
    import grid_oracle as GO
    from gfdl_atmos_cubed_sphere_amd.lib import cube_table
    npx = 13
    ref = GO.ref_sphere(npx)
    for kind in ("A", "B", "D", "C", "Dedge"):
        rt = ref.table(kind)
        for t in range(6):
            for m in range(len(rt[t])):
                a, b = rt[t][m], cube_table(prod, npx, kind, m, t)
                oa, ob = np.argsort(a["dst"], kind="stable"), np.argsort(b["dst"], kind="stable")
                for k in ("dst", "tile", "comp", "src") + (("sign",) if kind in ("D", "C", "Dedge") else ()):
                    assert np.array_equal(a[k][oa], b[k][ob]), (kind, t, m, k)


def test_cube_edge_exchange_through_rccl_loopback(prod):
    """EVERY cube-edge message of a C24 sphere through RCCL -- fv3_cube_halo_start / _complete with the six faces
    on this one GPU (24 + 24 grouped ncclSend / ncclRecv to the same rank per call, matched in posting order) -- against the oracle's
    update of the six tiles (= the device-gather result): every field kind, SCALAR_PAIR, mpp_get_boundary, a multi-field group"""
    from gfdl_atmos_cubed_sphere_amd.cubed_halo import CubeHalo, CubeHaloNative
    from gfdl_atmos_cubed_sphere_amd.lib import Context
    npx, npz = 25, 5
    cs, gs = PC.CC.sphere(npx)
    ctxs = [Context(g, npz, lib=prod) for g in gs]
    try:
        H = CubeHaloNative(ctxs, range(6), [0] * 6)
        G = CubeHalo(ctxs, npx, topo=PC.CC.product_topo(npx))
        rng = np.random.default_rng(0)
        bd = gs[0].bd
        mk = lambda k, nk=npz: [np.asfortranarray(rng.uniform(-1, 1, bd.shape(k, nk))) for _ in range(6)]      # noqa: E731
        for kind, kinds, vector in (("A", ("A",), True), ("B", ("B",), True), ("D", ("U", "V"), True), ("C", ("V", "U"), True),
                                    ("C", ("V", "U"), False), ("Dedge", ("U", "V"), True)):
            host = [mk(k) for k in kinds]
            dev = [[ctxs[t].from_host(a[t]) for t in range(6)] for a in host]
            dev2 = [[ctxs[t].from_host(a[t]) for t in range(6)] for a in host]
            ref = [[x.copy(order="F") for x in a] for a in host]
            cs.topo.update(kind, ref[0] if len(kinds) == 1 else (ref[0], ref[1]), vector=vector)
            H.update(kind, dev[0] if len(kinds) == 1 else (dev[0], dev[1]), vector=vector)
            G.update(kind, dev2[0] if len(kinds) == 1 else (dev2[0], dev2[1]), vector=vector)
            for m in range(len(kinds)):
                for t in range(6):
                    got = dev[m][t].download()
                    assert np.array_equal(got, ref[m][t]), (kind, vector, m, t)
                    assert np.array_equal(got, dev2[m][t].download()), ("gather", kind, vector, m, t)
        a1, a2, b1, u, v = mk("A"), mk("A", 1), mk("B"), mk("U"), mk("V")
        dev = {n: [ctxs[t].from_host(x[t]) for t in range(6)] for n, x in dict(a1=a1, a2=a2, b1=b1, u=u, v=v).items()}
        cs.topo.update("A", a1); cs.topo.update("A", a2); cs.topo.update("B", b1); cs.topo.update("D", (u, v))
        H.start([("A", dev["a1"]), ("A", dev["a2"]), ("B", dev["b1"]), ("D", (dev["u"], dev["v"]))])
        H.finish()
        for n, x in dict(a1=a1, a2=a2, b1=b1, u=u, v=v).items():
            for t in range(6):
                assert np.array_equal(dev[n][t].download(), x[t]), (n, t)
        G.close()
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("hydrostatic", [True, False])
def test_sphere_step_through_the_cube_edge_exchange_over_rccl(prod, hydrostatic):
    """a whole fv_dynamics call on the six faces with every halo update as RCCL messages behind the C ABI (loopback to this rank):
    the six-face oracle's state"""
    r = PC.check_jw_step(prod, npx=25, npz=20, k_split=1, n_split=2, bdt=900.0, hydrostatic=hydrostatic, nq=2, native_halo=True)
    assert r.pop("finite") == 1.0 and max(r.values()) <= 1e-12, r


def test_switched_off_paths_still_agree():
    """the forms the round-3 kernels replaced stay in the library behind switches read once per process: the column-kernel geopk, the
    pass chains of the damping operators and the LDS-tile transports of the damped levels, against the oracle, in their own process"""
    import subprocess, sys, textwrap
    here = os.path.dirname(os.path.abspath(__file__))
    code = textwrap.dedent("""
        import sys
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import parity_common as P, parity_nh as N, parity_cubed as PC
        from gfdl_atmos_cubed_sphere_amd import lib as L
        prod = L.load()
        N.check_halos_and_geopk(prod)
        PROD = dict(do_vort_damp=True, vtdm4=0.06, nord=3, d_con=1.0, dddmp=0.5)
        for hyd in (True, False):
            assert max(PC.check_d_sw(prod, npx=33, npz=17, hydrostatic=hyd, faces=(1,), flags=PROD, par_over=dict(dddmp=0.5)).values()) <= P.TOL
        print("ok")
    """) % (os.path.dirname(here), here)
    env = dict(os.environ, FV3_MI355X_GEOPK_PHASED="0", FV3_MI355X_DELN_FUSED="0", FV3_MI355X_FLUX_MARCH="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("hydrostatic", [False, True])
def test_face_group_one_launch_for_six_faces(prod, hydrostatic):
    """fv3_group: the six contexts of a sphere as one group -- every kernel the faces issue in turn is ONE launch
    with the face as the slowest grid index and the six functors by value in the kernel arguments -- against six separate launches
    per kernel: the same bits in every field, and nearly every launch of the substep loop ran all six faces at once.  C48 (pass
    kernels only), C96 with the marching interior, the production damping set."""
    PC.check_face_group(prod, npx=49, npz=8, hydrostatic=hydrostatic)
    PC.check_face_group(prod, npx=97, npz=6, n_split=2, hydrostatic=hydrostatic, flags=PROD)


@pytest.mark.parametrize("km", [8, 32, 79, 127])
def test_riem_lds_bit_identical_to_the_slab_kernels(prod, km):
    """the library's default Riemann solvers (dry, SIM1): levels across the lanes, pointwise work with the parity expressions, the
    recurrences in the reference's order -- sums and the pp system by hand-over rounds in registers, the stiff w system by one wavefront
    over the workgroup's 16 columns in LDS (csrc/nh_fast.h RiemFast<CG, true>) -- against the slab kernels (FV3_MI355X_RIEM_LDS=0):
    the same bits in every output (nh_utils.F90:1277-1394, nh_core.F90:47-241, nh_utils.F90:323-480)"""
    dims = dict(nx=200, ny=24, km=km) if km >= 79 else dict(nx=37, ny=13, km=km)     # ragged last 16-column block
    N.check_riem_lds_bits(prod, **dims)


@pytest.mark.parametrize("kw", [dict(), dict(hydrostatic=True, nq=0), dict(have_grid=True, consv_te=-2.0, tau=0.0), dict(what="dyn_core"),
                                dict(thermo=True), dict(thermo=True, what="dyn_core"),       # use_cond = moist_kappa = .true.
                                dict(do_diss_est=True), dict(fill_dp=True, do_diss_est=True, what="dyn_core"),    # flagstruct%do_diss_est, %fill_dp
                                dict(beta=-1.0)])                                                                 # one_grad_p (beta < -0.1)
def test_fortran_fv_dynamics_with_the_reference_argument_list_on_the_sphere(prod, tmp_path, kw):
    """VERDICT r3 item 6 (row a21): fv3_solo_refsig_sphere drives a C24 Jablonowski-Williamson fv_dynamics call with the REFERENCE'S
    argument list (one call per tile, host arrays with the fv_arrays layout, gridstruct / flagstruct / bd / domain), consv_te = 1,
    tau = 10 days, the virtual effect of the first tracer: six contexts in one process on this GPU (one launch group), the cube-edge
    exchange behind the C ABI -- bit-identical to FvDynamics.step_from_temperature on every tile (fv_dynamics.F90:79-936)"""
    import fortran_host as F
    if F.fortran_compiler() is None:
        pytest.skip("no Fortran compiler in this image")
    assert F.check_refsig_sphere(prod, tmp_path, npx=25, npz=20, n_split=2, k_split=2, bdt=900.0, **kw) == 0.0


@pytest.mark.parametrize("kw", [dict(), dict(hydrostatic=True, face_rank=(0, 0, 1, 1, 2, 2))])
def test_fortran_fv_dynamics_consv_am_on_the_sphere(prod, tmp_path, kw):
    """flagstruct%consv_am through the reference-signature fv_dynamics on the cubed sphere (fv_dynamics.F90:358-361, :747-800): compute_aam
    of every tile before and after the k_split loop, the two reproducing sums over the tiles (and the processes), u00, the wind correction
    -- against FvDynamics.step_from_temperature; the wrapper takes cos(lat) of gridstruct%agrid with the Fortran run-time's cos(), the
    Python host with numpy's, and u00 carries that last-bit difference into every wind: 1e-11, as on the doubly periodic domain"""
    import fortran_host as F
    if F.fortran_compiler() is None:
        pytest.skip("no Fortran compiler in this image")
    if "face_rank" in kw and True:
        kw = dict(kw, face_rank=(0, 0, 0, 0, 0, 0))     # one GPU here: the tiles of one process
    worst = F.check_refsig_sphere(prod, tmp_path, npx=25, npz=20, n_split=2, k_split=2, bdt=900.0, nq=0, consv_am=True, have_grid=True, tol=1e-11, **kw)
    assert worst <= 1e-11


@pytest.mark.parametrize("kw", [dict(nx=33, ny=9, km=20), dict(nx=200, ny=24, km=79), dict(nx=200, ny=24, km=127), dict(km=3), dict(km=8),
                                dict(km=40, lev_over=dict(do_vort_damp=True, vtdm4=0.06, nord=2))])
def test_edge_profile_lds_bit_identical_to_the_slab_kernel(prod, kw):
    """update_dz_d's edge_profile with the levels across the lanes and the elimination in the reference's order by hand-over rounds
    (csrc/nh_fast.h EdgeProfileLds, the library's default; the quotients through the host's correctly rounded reciprocals and a
    Markstein correction) against the slab kernel with its IEEE divisions (FV3_MI355X_RIEM_LDS=0): the same bits in zh and ws
    (nh_utils.F90:1590-1696, :204-320)"""
    N.check_edge_profile_lds_bits(prod, **kw)


@pytest.mark.parametrize("dims", [dict(nx=37, ny=13, km=32), dict(nx=200, ny=24, km=79), dict(nx=200, ny=24, km=127), dict(km=3)])
def test_riem_solvers_sim3_sim3p0_rim_2d(prod, dims):
    """the other vertical solvers Riem_Solver3 / Riem_Solver_c dispatch on a_imp (nh_core.F90:169-177, nh_utils.F90:449-459):
    SIM3p0_solver (a_imp < -0.999; C grid < -0.01, nh_utils.F90:1134-1274), SIM3_solver (a_imp < -0.5, :984-1132), RIM_2D (a_imp <= 0.5,
    :751-982) with one, four (the one-step branch of the layers the sound wave does not cross) and ten sub-steps -- against the oracle"""
    for a_imp, ms in ((-1.0, 1), (-0.75, 1), (0.3, 1), (0.3, 4), (0.0, 10), (-0.3, 3)):
        assert N.check_riem_solver3(prod, a_imp=a_imp, m_split=ms, use_logp=True, last_call=True, fp_out=True, **dims) <= 1e-13
        assert N.check_riem_solver3(prod, a_imp=a_imp, m_split=ms, last_call=False, **dims) <= 1e-13
        assert N.check_riem_solver_c(prod, a_imp=a_imp, m_split=ms, **dims) <= 1e-13


def test_fortran_fv_dynamics_reference_argument_list_with_the_energy_fixer_and_rayleigh_friction(prod, tmp_path):
    """fv_dynamics with the reference's argument list on the doubly periodic domain with consv_te = 1 and tau = 10 days carried in Fortran
    (fv3_fv_dynamics_call): bit-identical to FvDynamics.step_from_temperature on the GPU"""
    import fortran_host as F
    if F.fortran_compiler() is None:
        pytest.skip("no Fortran compiler in this image")
    F.check_fortran_fv_dynamics(prod, tmp_path, consv_te=1.0, tau=10.0, npz=16)
    F.check_fortran_fv_dynamics(prod, tmp_path, consv_te=-2.0, hydrostatic=True)


@pytest.mark.gpu
@pytest.mark.parametrize("hydrostatic", [False, True])
@pytest.mark.parametrize("flags", [None, dict(prevent_diss_cooling=False)])
def test_sponge_levels_run_on_the_marching_kernels(prod, hydrostatic, flags, monkeypatch):
    """levels 1, 2 of the default coefficients (del-2 damping of the divergence and of w, the heating of the latter) inside the branch-free
    marching kernels on a uniform grid: the oracle's values, and no LDS-tile launch left"""
    worst, rep = P.check_sponge_levels_march(prod, hydrostatic=hydrostatic, flags=flags)
    assert not ({"d_sw_transport", "d_sw_momentum", "d_sw_courant"} & set(rep)), rep
    # ... and the switch back to the tile kernels gives the same parity
    monkeypatch.setenv("FV3_MI355X_SPONGE_MARCH", "0")
    worst0, rep0 = P.check_sponge_levels_march(prod, hydrostatic=hydrostatic, flags=flags)
    assert "d_sw_transport" in rep0 and "d_sw_momentum" in rep0, rep0


@pytest.mark.gpu
def test_sponge_levels_march_smagorinsky_term(prod):
    """dddmp > 0 keeps the nord = 1 levels off the marching momentum kernel; the sponge form carries the term (sw_core.F90:1367) -- checked
    through the transport half, which still marches its sponge levels"""
    P.check_d_sw(prod, nx=70, ny=30, npz=4, perturb=False, par_over=dict(dddmp=0.2))


@pytest.mark.gpu
def test_mixed_segmentation_of_the_marching_launches(prod, monkeypatch):
    """balance_segments (tp2d_march.h): the first levels of a launch cut into one segment less than the others, so that the launch fills
    whole rounds of the chip.  FV3_MI355X_ROUND_SIMDS=35 makes that happen at test size in c_sw, the fused transport and the fused momentum
    kernel (70 x 60 x 9: the first 3 level slots in 7 segments, the others in 8); the oracle's values either way"""
    monkeypatch.setenv("FV3_MI355X_ROUND_SIMDS", "35")
    P.check_c_sw(prod, nx=70, ny=60, npz=9, perturb=False)
    P.check_d_sw(prod, nx=70, ny=60, npz=9, perturb=False)
    P.check_d_sw(prod, nx=70, ny=60, npz=9, perturb=False, hydrostatic=True)


@pytest.mark.gpu
@pytest.mark.parametrize("direction", ["x", "y"])
@pytest.mark.parametrize("iord", [5, -5, 6, 8])
def test_golden_ppm_lines_through_fv_tp_2d(prod, iord, direction):
    """the reference-held PPM vectors (tests/golden/, from the reference's own tp_core.ipynb) straight through the library's fv_tp_2d"""
    P.check_golden_ppm_through_fv_tp_2d(prod, iord, direction)


@pytest.mark.gpu
@pytest.mark.parametrize("which", [0, 1, 2])
@pytest.mark.parametrize("iord", [5, -5, 6, 8, 10])
def test_golden_ppm_lines_through_the_hip_operators(prod, iord, which):
    """all 144 reference-held PPM vectors (hord 10 among them) through the library's three 1-D operators: fv3_ppm_line"""
    P.check_golden_ppm_lines(prod, iord, which)


@pytest.mark.gpu
def test_config5_full_gnomonic_c768l79_face_33_tracers_properties(prod):
    """BASELINE configs[4] at FULL size on one GPU: one gnomonic C768 L79 face (768 x 768 x 79, grid_type 0, every metric row read), the
    hydrostatic c_sw + d_sw pair and one tracer_2d step of 33 tracers.  Past the oracle's reach, and a single face has no neighbours to
    close its mass budget with, so the properties are the local ones: every output finite; the flux-form update of delp IS the
    divergence of the mass fluxes the call accumulated (sw_core.F90:1059-1060, :928-940) on the whole compute domain; the monotone
    tracer scheme (hord_tr = 8) keeps every tracer inside the range it started with (fv_tracer2d.F90:471-541)."""
    import numpy as np
    from fields import smooth_state
    from gfdl_atmos_cubed_sphere_amd.cubed_sphere import CubedSphere
    from gfdl_atmos_cubed_sphere_amd.layout import Bounds
    from gfdl_atmos_cubed_sphere_amd.lib import Context
    from gfdl_atmos_cubed_sphere_amd.tracer2d import tracer_2d
    from test_oracle_properties import default_levels
    nx, npz, nq = 768, 79, 33
    g = CubedSphere(nx + 1).gridstruct(0)
    bd = g.bd
    ctx = Context(g, npz, lib=prod)
    try:
        st = smooth_state(Bounds(1, nx, 1, nx), npz, noise=0.05, hydrostatic=True)
        d = {k: ctx.from_host(v) for k, v in st.items()}
        for n, kind in P.CSW_OUT:
            d[n] = ctx.zeros(kind, npz)
        for n, kind in (("mfx", "FX"), ("mfy", "FY"), ("cx", "CX"), ("cy", "CY"), ("crx", "CX"), ("cry", "CY"),
                        ("xfx", "CX"), ("yfx", "CY"), ("delp_out", "A"), ("pt_out", "A"), ("u_out", "U"), ("v_out", "V")):
            d[n] = ctx.zeros(kind, npz)
        ctx.dsw_levels(default_levels(npz))
        par = dict(P.DSW_PAR)
        par.update(dt=12.5, hydrostatic=1, use_cond=0)      # C768: dt_atmos 150 s / k_split 2 / n_split 6
        dt = par["dt"]
        ctx.c_sw(d["delpc"], d["delp"], d["ptc"], d["pt"], d["u"], d["v"], None, d["uc"], d["vc"], d["ua"], d["va"], None, d["ut"],
                 d["vt"], d["divg_d"], 1, 0.5 * dt, True)
        ctx.d_sw(par, None, d["delp"], d["pt"], d["u"], d["v"], None, d["uc"], d["vc"], d["ua"], d["va"], d["divg_d"], d["mfx"], d["mfy"],
                 d["cx"], d["cy"], d["crx"], d["cry"], d["xfx"], d["yfx"], None, d["delp_out"], d["pt_out"], d["u_out"], d["v_out"], None,
                 None, None, None)
        r = (bd.is_, bd.ie, bd.js, bd.je)
        dp0 = bd.view(st["delp"], "A", *r)
        dp1 = bd.view(d["delp_out"].download(), "A", *r)
        mfx, mfy = d["mfx"].download(), d["mfy"].download()
        div = (mfx[:-1] - mfx[1:] + mfy[:, :-1] - mfy[:, 1:]) * bd.view(g.rarea, "A", *r)[:, :, None]
        assert np.all(np.isfinite(dp1)) and np.max(np.abs((dp1 - dp0) - div)) <= 1e-12 * np.max(np.abs(dp0))
        for n in ("pt_out", "u_out", "v_out"):
            assert np.all(np.isfinite(bd.view(d[n].download(), {"pt_out": "A", "u_out": "U", "v_out": "V"}[n], *r))), n
        # 33 tracers through one tracer_2d call with the fluxes / Courant numbers this substep accumulated
        rng = np.random.default_rng(5)
        q0 = np.asfortranarray(rng.uniform(0.0, 1.0, bd.shape("A", npz) + (nq,)))

        class NoHalo:      # one face alone: the halo of q holds its initial values (inside the range as well); one sub-cycle (q_split = 1)
            world = 1

            def update(self, fields):
                pass
        dq, dqn, dpn = ctx.from_host(q0), ctx.zeros("A", npz * nq), ctx.zeros("A", npz)
        qr, _, nsplt = tracer_2d(ctx, NoHalo(), dq, dqn, d["delp"], dpn, d["mfx"], d["mfy"], d["cx"], d["cy"], d["xfx"], d["yfx"], nq, hord=8,
                                 q_split=1)
        q1 = qr.download().reshape(q0.shape, order="F")
        inner = q1[bd.ng:bd.ng + nx, bd.ng:bd.ng + nx]
        assert nsplt == 1 and np.all(np.isfinite(inner))
        lo, hi = float(q0.min()), float(q0.max())
        assert inner.min() >= lo - 1e-12 and inner.max() <= hi + 1e-12, (inner.min(), inner.max())
    finally:
        ctx.close()
