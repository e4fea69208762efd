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
