"""The cubed sphere on the CPU: topology tables against the geometry, the metric terms, and the oracle's grid_type < 3
branches of c_sw / d_sw / fv_tp_2d / xtp_u / ytp_v driven over the six faces with emulated halo updates.  The reference
ships no golden data for these branches ("parity unpinned"); what pins them here are the identities the reference's design
guarantees: the two faces of an edge compute the same edge quantity (to rounding), so air mass and tracer mass are
conserved globally to rounding, and a halo value equals the neighbour's interior value."""
import numpy as np
import pytest

import cubed_common as CC
from gfdl_atmos_cubed_sphere_amd.cubed_sphere import RADIUS, CubedSphere, _mid, _unit

F = np.asfortranarray


def test_topology_rotates_vectors_like_the_geometry():
    npx, ng = 9, 3
    cs = CubedSphere(npx)
    topo = cs.topo
    U, V, UC, VC = [], [], [], []
    for t in range(6):
        g3 = cs.grids[t]["grid3"]
        tx, mx = _unit(g3[1:, :] - g3[:-1, :]), _mid(g3[1:, :], g3[:-1, :])
        ty, my = _unit(g3[:, 1:] - g3[:, :-1]), _mid(g3[:, 1:], g3[:, :-1])
        U.append(F(np.sum(CC.wind(mx) * tx, -1)))
        V.append(F(np.sum(CC.wind(my) * ty, -1)))
        UC.append(F(np.sum(CC.wind(my) * np.cross(ty, my), -1)))     # normal to the y-edges, towards +x
        VC.append(F(np.sum(CC.wind(mx) * np.cross(mx, tx), -1)))     # normal to the x-edges, towards +y
    for kind, true in (("D", (U, V)), ("C", (UC, VC))):
        work = tuple([x.copy(order="F") for x in lst] for lst in true)
        tab = topo.table(kind)
        for m in range(2):
            for t in range(6):
                f = work[m][t].reshape(-1, order="F")
                f[tab[t][m]["dst"]] = np.nan
                work[m][t][...] = f.reshape(work[m][t].shape, order="F")
        topo.update(kind, work)
        for m in range(2):
            for t in range(6):
                d = tab[t][m]["dst"]
                got, ref = work[m][t].reshape(-1, order="F")[d], true[m][t].reshape(-1, order="F")[d]
                assert np.max(np.abs(got - ref)) < 1e-13 * 50.0, (kind, m, t)


def test_metric_terms():
    npx, ng = 13, 3
    cs = CubedSphere(npx)
    s = slice(ng, ng + npx - 1)
    total = sum(g["area"][s, s].sum() for g in cs.grids)
    assert abs(total / (4 * np.pi * RADIUS ** 2) - 1.0) < 1e-13
    g0 = cs.grids[0]
    for t in (1, 2, 3, 4, 5):                       # the six faces are congruent
        for k in ("area", "dxa", "dya"):
            assert np.max(np.abs(cs.grids[t][k][s, s] - g0[k][s, s])) < 1e-12 * g0[k][s, s].max()
    gs = cs.gridstruct(0)
    sc = slice(ng, ng + npx)
    assert 0.4 < gs.m["sina"][sc, sc].min() <= gs.m["sina"][sc, sc].max() <= 1.0
    assert np.all(np.isfinite(gs.m["edge_w"][1:-1])) and np.all((gs.m["edge_w"][1:-1] > 0.2) & (gs.m["edge_w"][1:-1] < 0.8))
    # halo geometry = the neighbour's geometry: the spacing across a face edge continues the interior spacing
    from gfdl_atmos_cubed_sphere_amd.cubed_sphere import _gcd3
    for t in range(6):
        g3 = cs.grids[t]["grid3"]
        r = _gcd3(g3[ng - 1, sc], g3[ng, sc]) / _gcd3(g3[ng, sc], g3[ng + 1, sc])
        assert np.max(np.abs(r - 1.0)) < 1e-12


@pytest.mark.parametrize("hydrostatic,par,flags", [
    (False, None, None),
    (True, None, None),
    (False, dict(hord_mt=5, hord_vt=5, hord_tm=5, hord_dp=5), None),
    (False, dict(hord_mt=6, hord_vt=6, hord_tm=6, hord_dp=-5), None),
    (True, dict(hord_mt=8, hord_vt=9, hord_tm=8, hord_dp=8), None),
    (False, dict(dddmp=0.2), dict(nord=2, dddmp=0.2)),
    (False, None, dict(nord=3)),
])
def test_oracle_pair_conserves_mass_on_the_sphere(hydrostatic, par, flags):
    npx, npz, ng = 13, 3, 3
    cs, gs, before, after = CC.oracle_pair(npx, npz, dt=600.0, hydrostatic=hydrostatic, par_over=par, flags=flags)
    s = slice(ng, ng + npx - 1)
    m0 = sum((b["delp"][s, s, :] * g.m["area"][s, s, None]).sum() for b, g in zip(before, gs))
    m1 = sum((a["delp"][s, s, :] * g.m["area"][s, s, None]).sum() for a, g in zip(after, gs))
    assert abs(m1 - m0) <= 1e-14 * abs(m0), (m1 - m0) / m0
    t0 = sum((b["pt"][s, s, :] * b["delp"][s, s, :] * g.m["area"][s, s, None]).sum() for b, g in zip(before, gs))
    t1 = sum((a["pt"][s, s, :] * a["delp"][s, s, :] * g.m["area"][s, s, None]).sum() for a, g in zip(after, gs))
    assert abs(t1 - t0) <= 1e-13 * abs(t0), (t1 - t0) / t0
    for a, b in zip(after, before):
        for n in ("delp", "pt", "u", "v") + (() if hydrostatic else ("w",)):
            assert np.all(np.isfinite(a[n][s, s, :]))
        assert np.max(np.abs(a["delp"][s, s, :] - b["delp"][s, s, :])) > 1e-3          # the step did something
    # the two faces of contact 1 (face 1 east = face 2 west) computed the same edge fluxes
    N = npx - 1
    e, w = after[0]["mfx"][N, :, :], after[1]["mfx"][0, :, :]
    assert np.max(np.abs(e - w)) <= 1e-13 * np.max(np.abs(e))


def _divergent_state(npx, npz, seed=5):
    """a purely divergent grid-scale wind V = grad(chi) on the unit sphere as a function of POSITION (the two faces of an edge hold
    the same value), uniform delp / pt, no mean flow"""
    cs, gs = CC.sphere(npx)
    rng = np.random.default_rng(seed)
    kmax = 1.6 * (npx - 1)                  # up to ~ 2.5 dx waves (dx ~ pi / (2 (npx - 1)) rad)
    K = rng.normal(size=(24, 3))
    K = K / np.linalg.norm(K, axis=1, keepdims=True) * rng.uniform(0.4 * kmax, kmax, size=(24, 1))
    ph, amp = rng.uniform(0, 2 * np.pi, 24), rng.uniform(0.5, 1.0, 24)

    def V(p):
        g = np.zeros(p.shape)
        for k, f, a in zip(K, ph, amp):
            g += (a / np.linalg.norm(k)) * np.cos(p @ k + f)[..., None] * k
        return 3.0 * (g - np.sum(g * p, -1, keepdims=True) * p)       # tangential part of the 3-D gradient

    st = []
    for t in range(6):
        g3, a3 = cs.grids[t]["grid3"], cs.grids[t]["agrid3"]
        tx, mx = CC._unit(g3[1:, :] - g3[:-1, :]), CC._mid(g3[1:, :], g3[:-1, :])
        ty, my = CC._unit(g3[:, 1:] - g3[:, :-1]), CC._mid(g3[:, 1:], g3[:, :-1])
        u, v = np.sum(V(mx) * tx, -1), np.sum(V(my) * ty, -1)
        one = np.ones(a3.shape[:2] + (npz,))
        st.append(dict(u=F(np.repeat(u[..., None], npz, 2)), v=F(np.repeat(v[..., None], npz, 2)), delp=F(800.0 * one), pt=F(300.0 * one)))
    CC.exchange_pair(cs, st, "u", "v", "D")
    return cs, gs, st


@pytest.mark.parametrize("nord,d4_bg", [(1, 0.16), (1, 0.05), (2, 0.12), (2, 0.15), (3, 0.12), (3, 0.15)])
def test_divergence_damping_dissipates_at_the_cube_corners_too(nord, d4_bg):
    """VERDICT r2 item 1c.  One c_sw -> d_sw with dt -> 0 applies only the del-2(nord+1) divergence damping (its coefficient
    (d4_bg da_min_c)^(nord+1) does not scale with dt, sw_core.F90:1449-1460; every flux term does).  On a purely divergent grid-scale
    wind the increment must remove kinetic energy globally AND around the eight cube corners, with a strength there comparable to
    the face interior.  With the round-2 corner area_c (da_min_c 2.6 x too small, rarea_c 2.6 x too large at the corner point) the
    corner ratio was 7 - 49 x off and the corner increment had the wrong sign for nord >= 2."""
    npx, npz, ng = 25, 2, 3
    cs, gs, st = _divergent_state(npx, npz)
    N = npx - 1
    _, _, before, after = CC.oracle_pair(npx, npz, dt=1.0e-6, hydrostatic=True, par_over=dict(d4_bg=d4_bg),
                                         flags=dict(nord=nord, d4_bg=d4_bg, n_sponge=-1, d2_bg=0.0), st=st)
    su, sv = (slice(ng, ng + N), slice(ng, ng + N + 1)), (slice(ng, ng + N + 1), slice(ng, ng + N))
    ke0 = ke1 = 0.0
    ck0 = ck1 = ik0 = ik1 = 0.0
    w = 3                                    # "around a corner": the 3 x 3 edges next to it
    for t in range(6):
        g = gs[t].m
        u0, v0 = before[t]["u"][su][..., 0], before[t]["v"][sv][..., 0]
        u1, v1 = after[t]["u"][su][..., 0] * g["rdx"][su], after[t]["v"][sv][..., 0] * g["rdy"][sv]     # d_sw returns u dx, v dy
        wu, wv = (g["dx"] * g["dyc"])[su].copy(), (g["dy"] * g["dxc"])[sv].copy()
        wu[:, 0] *= 0.5; wu[:, -1] *= 0.5; wv[0, :] *= 0.5; wv[-1, :] *= 0.5          # shared edges are held by two faces
        ke0 += (wu * u0 ** 2).sum() + (wv * v0 ** 2).sum()
        ke1 += (wu * u1 ** 2).sum() + (wv * v1 ** 2).sum()
        for ci in (slice(0, w), slice(N - w, N)):
            for cj in (slice(0, w), slice(N - w, N)):
                ck0 += (wu * u0 ** 2)[ci, cj].sum() + (wv * v0 ** 2)[ci, cj].sum()
                ck1 += (wu * u1 ** 2)[ci, cj].sum() + (wv * v1 ** 2)[ci, cj].sum()
        m = slice(N // 2 - 4, N // 2 + 4)
        ik0 += (wu * u0 ** 2)[m, m].sum() + (wv * v0 ** 2)[m, m].sum()
        ik1 += (wu * u1 ** 2)[m, m].sum() + (wv * v1 ** 2)[m, m].sum()
    assert ke1 < ke0, (nord, d4_bg, ke1 / ke0)
    assert ck1 < ck0, ("kinetic energy grows around the cube corners", nord, d4_bg, ck1 / ck0)
    rc, ri = 1.0 - ck1 / ck0, 1.0 - ik1 / ik0
    assert ri > 0.0 and 0.25 < rc / ri < 4.0, ("corner / interior damping rate", nord, d4_bg, rc, ri)
