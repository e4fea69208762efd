"""Initial conditions and vertical coordinates of the BASELINE configurations (host-side set-up code, numpy).

``set_eta``: the hybrid sigma-pressure levels ``ak, bk`` of tools/fv_eta.F90:272-811 for npz = 79 (``var_hi``, :1166-1341,
ptop = 1 Pa, stretch 1.03) and npz = 127 (``var_gfs``, :1002-1164, ptop = 1 Pa, pint = 75 hPa, stretch 1.028).
``jablonowski_williamson``: test_case = 13, the baroclinic-wave initial condition of tools/test_cases.F90:1575-1890
(adiabatic: q = 0; the Gaussian zonal-wind perturbation of radius a/10 centred on (20 E, 40 N)).
"""
from __future__ import annotations

import numpy as np

from .cubed_sphere import OMEGA, RADIUS, CubedSphere, _mid, _unit, gc_dist, latlon_of
from .lib import GRAV, KAPPA, RDGAS


def _sm1_edge(ze, ntimes):
    """sm1_edge (fv_eta.F90:2313-2346): ntimes passes of a 1-2-1 smoother on the layer thicknesses"""
    km = ze.size - 1
    df = 0.25
    dz = ze[1:] - ze[:-1]              # dz(k) = ze(k+1) - ze(k), k = 1..km
    k2 = km - 1
    for n in range(1, ntimes + 1):
        k1 = 2 + (ntimes - n)
        flux = np.zeros(km + 2)
        for k in range(k1 + 1, k2 + 1):
            flux[k] = df * (dz[k - 1] - dz[k - 2])
        for k in range(k1, k2 + 1):
            dz[k - 1] = dz[k - 1] - flux[k] + flux[k + 1]
    out = ze.copy()
    for k in range(km, 0, -1):
        out[k - 1] = out[k] - dz[k - 1]
    return out


def _levels_from_stretch(km, ptop, pint, s_fac, smooth):
    p00, t0 = 1.0e5, 270.0
    peln1, pelnb = np.log(ptop), np.log(p00)
    ztop = RDGAS / GRAV * t0 * (pelnb - peln1)
    dz = s_fac * (ztop / np.sum(s_fac))
    ze = np.zeros(km + 1)
    for k in range(km - 1, -1, -1):
        ze[k] = ze[k + 1] + dz[k]
    dz = dz * (ztop / ze[0])            # re-scale dz with the stretched ztop
    for k in range(km - 1, -1, -1):
        ze[k] = ze[k + 1] + dz[k]
    if smooth:
        ze = _sm1_edge(ze, smooth)
    dz = ze[:-1] - ze[1:]
    dlnp = GRAV * dz / (RDGAS * t0)
    peln = np.empty(km + 1)
    pe1 = np.empty(km + 1)
    peln[0], pe1[0], peln[km], pe1[km] = peln1, ptop, pelnb, p00
    for k in range(1, km):
        peln[k] = peln[k - 1] + dlnp[k - 1]
        pe1[k] = np.exp(peln[k])
    ks = 0
    for k in range(2, km + 1):           # Fortran k = 2..km
        if pint < pe1[k - 1]:
            ks = k - 1
            break
    eta = pe1 / pe1[km]
    ep, es = eta[ks], eta[km - 1]        # eta(ks+1), eta(km)
    alpha = (ep ** 2 - 2.0 * ep * es) / (es - ep) ** 2
    beta = 2.0 * ep * es ** 2 / (es - ep) ** 2
    gama = -(ep * es) ** 2 / (es - ep) ** 2
    ak, bk = np.zeros(km + 1), np.zeros(km + 1)
    ak[:ks + 1] = eta[:ks + 1] * 1.0e5
    for k in range(ks + 1, km):          # Fortran ks+2..km
        ak[k] = (alpha * eta[k] + beta + gama / eta[k]) * 1.0e5
        bk[k] = (pe1[k] - ak[k]) / pe1[km]
    ak[km], bk[km] = 0.0, 1.0
    return ak, bk, ks


def set_eta(km: int):
    """ak, bk, ks, ptop of the reference's default level sets for km = 79 (var_hi) and km = 127 (var_gfs)"""
    s_fac = np.zeros(km)
    if km == 79:                          # fv_eta.F90:656-666 -> var_hi
        ptop, pint, s_rate, k_inc, s0 = 1.0, 100.0e2, 1.03, 15, 0.10
        s_inc = (1.0 - s0) / k_inc
        s_fac[km - 1] = s0
        for k in range(km - 1, km - k_inc - 1, -1):        # Fortran k = km-1 .. km-k_inc
            s_fac[k - 1] = s_fac[k] + s_inc
        s_fac[km - k_inc - 2] = 0.5 * (s_fac[km - k_inc - 1] + s_rate)
        for k in range(km - k_inc - 2, 8, -1):             # Fortran k = km-k_inc-2 .. 9
            s_fac[k - 1] = s_rate * s_fac[k]
        s_fac[7] = 0.5 * (1.1 + s_rate) * s_fac[8]
        s_fac[6] = 1.1 * s_fac[7]
        s_fac[5] = 1.15 * s_fac[6]
        s_fac[4] = 1.2 * s_fac[5]
        s_fac[3] = 1.3 * s_fac[4]
        s_fac[2] = 1.4 * s_fac[3]
        s_fac[1] = 1.45 * s_fac[2]
        s_fac[0] = 1.5 * s_fac[1]
        smooth = 1
    elif km == 127:                       # fv_eta.F90:726-745 -> var_gfs
        ptop, pint, s_rate, k_inc, s0 = 1.0, 75.0e2, 1.028, 25, 0.13
        s_inc = (1.0 - s0) / k_inc
        s_fac[km - 1] = s0
        for k in range(km - 1, km - k_inc - 1, -1):
            s_fac[k - 1] = s_fac[k] + s_inc
        for k in range(km - k_inc - 1, 8, -1):             # Fortran k = km-k_inc-1 .. 9
            s_fac[k - 1] = s_rate * s_fac[k]
        s_fac[7] = 0.5 * (1.1 + s_rate) * s_fac[8]
        s_fac[6] = 1.10 * s_fac[7]
        s_fac[5] = 1.15 * s_fac[6]
        s_fac[4] = 1.20 * s_fac[5]
        s_fac[3] = 1.26 * s_fac[4]
        s_fac[2] = 1.33 * s_fac[3]
        s_fac[1] = 1.41 * s_fac[2]
        s_fac[0] = 1.60 * s_fac[1]
        smooth = 0
    else:
        raise ValueError("set_eta: the level sets built are npz = 79 and npz = 127")
    ak, bk, ks = _levels_from_stretch(km, ptop, pint, s_fac, smooth)
    return ak, bk, ks, ak[0]


def _east_component(tangent, p):
    """tangent . (unit vector to the east at p) = e(2) cos(lon) - e(1) sin(lon)  (test_cases.F90:1661)"""
    lon, _ = latlon_of(p)
    return tangent[..., 1] * np.cos(lon) - tangent[..., 0] * np.sin(lon)


def jablonowski_williamson(cs: CubedSphere, ak, bk, hydrostatic: bool = True, perturb: bool = True):
    """test_case = 13 on every face: dict(u, v, delp, pt (temperature), phis[, w, delz]) per face, compute domain filled
    (halos are the caller's exchange).  Arrays have the reference's shapes incl. halo."""
    ak, bk = np.asarray(ak, dtype=np.float64), np.asarray(bk, dtype=np.float64)
    npz, npx, ng = ak.size - 1, cs.npx, cs.ng
    N = npx - 1
    R, om = cs.radius, cs.omega
    F = np.asfortranarray
    eta_0, Ubar = 0.252, 35.0
    eta = 0.5 * ((ak[:-1] + ak[1:]) / 1.0e5 + bk[:-1] + bk[1:])
    eta_v = (eta - eta_0) * np.pi * 0.5
    pcen = (np.pi / 9.0, 2.0 * np.pi / 9.0)
    u1, r0 = (1.0, R / 10.0) if perturb else (0.0, 1.0)
    T_0, delta_T, lapse, eta_t, eta_s = 288.0, 480000.0, 0.005, 0.2, 1.0

    # level-independent factors are evaluated once per point set (the products keep the association of the level-wise forms)
    def uzonal_parts(p):
        lon, lat = latlon_of(p)
        r = gc_dist(pcen[0], pcen[1], lon, lat, R)
        arg = -(r / r0) ** 2.0
        return np.sin(2.0 * lat) ** 2.0, np.where(arg > -40.0, u1 * np.exp(np.maximum(arg, -40.0)), 0.0)

    def uzonal(parts, z):
        return Ubar * np.cos(eta_v[z]) ** 1.5 * parts[0] + parts[1]

    def t_parts(lat):
        return ((-2.0 * (np.sin(lat) ** 6.0) * (np.cos(lat) ** 2.0 + 1.0 / 3.0) + 10.0 / 63.0) * 2.0 * Ubar,
                ((8.0 / 5.0) * (np.cos(lat) ** 3.0) * (np.sin(lat) ** 2.0 + 2.0 / 3.0) - np.pi / 4.0) * R * om)

    def t_of(parts, z, t_mean):
        return t_mean + 0.75 * (eta[z] * np.pi * Ubar / RDGAS) * np.sin(eta_v[z]) * np.sqrt(np.cos(eta_v[z])) * (
            parts[0] * np.cos(eta_v[z]) ** 1.5 + parts[1])

    def phis_of(lat):
        c = np.cos((eta_s - eta_0) * np.pi / 2.0)
        return Ubar * c ** 1.5 * ((-2.0 * (np.sin(lat) ** 6.0) * (np.cos(lat) ** 2.0 + 1.0 / 3.0) + 10.0 / 63.0) * Ubar * c ** 1.5 +
                                  ((8.0 / 5.0) * (np.cos(lat) ** 3.0) * (np.sin(lat) ** 2.0 + 2.0 / 3.0) - np.pi / 4.0) * R * om)

    out = []
    s = slice(ng, ng + N)          # cells 1..N
    sc = slice(ng, ng + N + 1)     # corners 1..npx
    nid = N + 2 * ng
    for t in range(6):
        g3 = cs.grids[t]["grid3"]
        a3 = cs.grids[t]["agrid3"]
        c = g3[sc, sc]                                                  # corners (npx, npx, 3)
        # tangent unit vectors at the corners (ee1, ee2: fv_grid_utils.F90:498-520; one-sided on the face edges) and at the
        # mid-points of the cell edges (es(:,:,:,1), ew(:,:,:,2): :247-323)
        gx = g3[ng - 1:ng + N + 2, sc]                                  # i = 0..npx+1
        gy = g3[sc, ng - 1:ng + N + 2]
        lo, hi = gx[:-2].copy(), gx[2:].copy()
        lo[0], hi[-1] = c[0], c[-1]                                     # i == 1: (i, i+1); i == npx: (i-1, i)
        ee1 = _unit(np.cross(np.cross(lo, hi), c))
        lo, hi = gy[:, :-2].copy(), gy[:, 2:].copy()
        lo[:, 0], hi[:, -1] = c[:, 0], c[:, -1]
        ee2 = _unit(np.cross(np.cross(lo, hi), c))
        mx = _mid(c[:-1, :], c[1:, :])                                  # mid-points of the x-edges (N, npx)
        es1 = _unit(np.cross(np.cross(c[:-1, :], c[1:, :]), mx))
        my = _mid(c[:, :-1], c[:, 1:])                                  # y-edges (npx, N)
        ew2 = _unit(np.cross(np.cross(c[:, :-1], c[:, 1:]), my))
        u = np.zeros((nid, nid + 1, npz), order="F")
        v = np.zeros((nid + 1, nid, npz), order="F")
        pt = np.zeros((nid, nid, npz), order="F")
        lat_c = latlon_of(c)[1]
        lat_a = latlon_of(a3[s, s])[1]
        lat_mx, lat_my = latlon_of(mx)[1], latlon_of(my)[1]
        zp = [uzonal_parts(x) for x in (c[:-1, :], c[1:, :], mx, c[:, :-1], c[:, 1:], my)]
        ec = [_east_component(a, b) for a, b in ((ee1[:-1, :], c[:-1, :]), (ee1[1:, :], c[1:, :]), (es1, mx),
                                                 (ee2[:, :-1], c[:, :-1]), (ee2[:, 1:], c[:, 1:]), (ew2, my))]
        tp = [t_parts(x) for x in (lat_a, lat_mx, lat_my, lat_c)]
        for z in range(npz):
            uu1, uu3, uu2 = uzonal(zp[0], z) * ec[0], uzonal(zp[1], z) * ec[1], uzonal(zp[2], z) * ec[2]
            u[s, sc, z] = 0.25 * (uu1 + 2.0 * uu2 + uu3)
            vv3, vv1, vv2 = uzonal(zp[3], z) * ec[3], uzonal(zp[4], z) * ec[4], uzonal(zp[5], z) * ec[5]
            v[sc, s, z] = 0.25 * (vv1 + 2.0 * vv2 + vv3)
            t_mean = T_0 * eta[z] ** (RDGAS * lapse / GRAV)
            if eta_t > eta[z]:
                t_mean = t_mean + delta_T * (eta_t - eta[z]) ** 5.0
            pt1 = t_of(tp[0], z, t_mean)
            pe_ = t_of(tp[1], z, t_mean)       # x-edge mid-points: south (j) and north (j+1) edges of the cells
            pw_ = t_of(tp[2], z, t_mean)       # y-edge mid-points: west (i) and east (i+1)
            pc_ = t_of(tp[3], z, t_mean)
            pt[s, s, z] = (0.25 * pt1 + 0.125 * (pe_[:, :-1] + pw_[1:, :] + pe_[:, 1:] + pw_[:-1, :]) +
                           0.0625 * (pc_[:-1, :-1] + pc_[1:, :-1] + pc_[1:, 1:] + pc_[:-1, 1:]))
        ps = 1.0e5
        delp = np.zeros((nid, nid, npz), order="F")
        delp[s, s, :] = (ak[1:] - ak[:-1]) + ps * (bk[1:] - bk[:-1])
        pe_ = phis_of(lat_mx)
        pw_ = phis_of(lat_my)
        pc_ = phis_of(lat_c)
        phis = np.zeros((nid, nid), order="F")
        phis[s, s] = (0.25 * phis_of(lat_a) + 0.125 * (pe_[:, :-1] + pw_[1:, :] + pe_[:, 1:] + pw_[:-1, :]) +
                      0.0625 * (pc_[:-1, :-1] + pc_[1:, :-1] + pc_[1:, 1:] + pc_[:-1, 1:]))
        d = dict(u=F(u), v=F(v), delp=F(delp), pt=F(pt), phis=F(phis))
        if not hydrostatic:
            pe = ak[0] + np.concatenate([[0.0], np.cumsum(delp[ng, ng, :])])
            peln = np.log(pe)
            d["w"] = np.zeros((nid, nid, npz), order="F")
            d["delz"] = F(RDGAS / GRAV * pt[s, s, :] * (peln[:-1] - peln[1:])[None, None, :])
        out.append(d)
    return out


def supercell_sounding(pk1, ps=1.0e5, kappa=2.0 / 7.0, rdgas=287.04, grav=9.80, rvgas=461.50):
    """SuperCell_Sounding (tools/test_cases.F90:6500-6616): the Weisman & Klemp sounding on the model levels.  pk1: layer-mean
    p**kappa of the column; returns (temperature, specific humidity) per layer.  401 height levels of 50 m, three passes of the
    hydrostatic integration with the virtual effect, saturation by the routine's own formula (:6578)."""
    pk1 = np.asarray(pk1, dtype=np.float64)
    ns, nx_ = 401, 3
    cp_air = rdgas / kappa
    tmin, p00, qst, qv0, ztr, ttr, ptr, pt0 = 175.0, 1.0e5, 3.0e-6, 1.4e-2, 12.0e3, 213.0, 343.0, 300.0
    zvir = rvgas / rdgas - 1.0
    pk0 = p00 ** kappa
    zs = 50.0 * np.arange(ns - 1, -1, -1, dtype=np.float64)           # zs(ns) = 0 at the surface
    qs = np.full(ns, qst)
    rh = np.full(ns, 0.25)
    pt = np.empty(ns)
    strat = zs > ztr
    pt[strat] = ptr * np.exp(grav * (zs[strat] - ztr) / (cp_air * ttr))
    fac = (zs[~strat] / ztr) ** 1.25
    pt[~strat] = pt0 + (ptr - pt0) * fac
    rh[~strat] = 1.0 - 0.75 * fac
    qs[~strat] = qv0 - (qv0 - qst) * fac
    pt = pt / pk0
    pk = np.empty(ns)
    pk[-1] = ps ** kappa
    for _ in range(nx_):
        temp1 = 0.5 * (pt[:-1] * (1.0 + zvir * qs[:-1]) + pt[1:] * (1.0 + zvir * qs[1:]))
        dpk = grav * (zs[:-1] - zs[1:]) / (cp_air * temp1)
        for k in range(ns - 2, -1, -1):
            pk[k] = pk[k + 1] - dpk[k]
        if np.any(pk <= 0.0):
            raise FloatingPointError("Super-Cell case: pk < 0")
        t1 = pt * pk
        pp = np.exp(np.log(pk) / kappa)
        qs = np.minimum(qv0, rh * (380.0 / pp * np.exp(17.27 * (t1 - 273.0) / (t1 - 36.0))))
    tp, qp = np.empty_like(pk1), np.empty_like(pk1)
    for k, x in enumerate(pk1):
        if x <= pk[0]:
            tp[k], qp[k] = pt[0] * pk[0] / x, qst
        elif x >= pk[-1]:
            tp[k], qp[k] = pt[-1], qs[-1]
        else:
            kk = int(np.searchsorted(pk, x, side="left")) - 1
            kk = min(max(kk, 0), ns - 2)
            f = (x - pk[kk]) / (pk[kk + 1] - pk[kk])
            tp[k], qp[k] = pt[kk] + (pt[kk + 1] - pt[kk]) * f, qs[kk] + (qs[kk + 1] - qs[kk]) * f
    return np.maximum(tmin, tp * pk1), qp


def supercell(bd, npz, ak, bk, dx_const, dy_const, npx_global=None, npy_global=None, umean=25.0, bubble=True, dt_amp=2.0,
              dt_rad=10.0e3, kappa=2.0 / 7.0, rdgas=287.04, grav=9.80, rvgas=461.50):
    """test_case = 17 of init_double_periodic (tools/test_cases.F90:4966-5057): the doubly periodic supercell with a straight wind
    -- the Weisman-Klemp sounding on the levels, u = Umean*tanh(z/3 km) - Umean/2, v = w = 0, hydrostatic delz from the virtual
    temperature (p_var), a warm bubble of dt_amp K and radius dt_rad in the middle of the domain.  BASELINE config 4's initial
    condition.  (Umean is a namelist value the reference tree does not default; 25 m/s is the Weisman-Klemp wind.)  Returns the
    fields of this rank's block bd with halos unfilled: u, v, w, delp, pt (TEMPERATURE), delz, phis, q (A x npz x 1: vapour)."""
    ak, bk = np.asarray(ak, dtype=np.float64), np.asarray(bk, dtype=np.float64)
    ptop, p00 = float(ak[0]), 1000.0e2
    zvir = rvgas / rdgas - 1.0
    pe = ak + p00 * bk
    peln = np.log(pe)
    pk = np.exp(kappa * peln)
    pk1 = (pk[1:] - pk[:-1]) / (kappa * (peln[1:] - peln[:-1]))
    ts1, qs1 = supercell_sounding(pk1, p00, kappa, rdgas, grav, rvgas)
    dp = (ak[1:] - ak[:-1]) + p00 * (bk[1:] - bk[:-1])
    delz1 = (rdgas / grav) * ts1 * (1.0 + zvir * qs1) * (peln[:-1] - peln[1:])          # p_var, make_nh (negative)
    ze1 = np.concatenate([np.cumsum((-delz1)[::-1])[::-1], [0.0]])
    zm = 0.5 * (ze1[:-1] + ze1[1:])
    out = {"u": bd.zeros("U", npz), "v": bd.zeros("V", npz), "w": bd.zeros("A", npz), "delp": bd.zeros("A", npz),
           "pt": bd.zeros("A", npz), "delz": bd.zeros("CC", npz), "phis": bd.zeros("A"), "q": np.zeros(bd.shape("A", npz) + (1,), order="F")}
    out["u"][...] = (umean * np.tanh(zm / 3.0e3) - 0.5 * umean)[None, None, :]
    out["delp"][...] = dp[None, None, :]
    out["pt"][...] = ts1[None, None, :]
    out["delz"][...] = delz1[None, None, :]
    out["q"][..., 0] = qs1[None, None, :]
    if bubble:
        npx = npx_global if npx_global is not None else bd.nx + 1
        npy = npy_global if npy_global is not None else bd.ny + 1
        ic, jc, zc = (npx - 1) // 2 + 1, (npy - 1) // 2 + 1, 1.4e3
        ii = np.arange(bd.is_ - bd.ng, bd.ie + bd.ng + 1, dtype=np.float64)[:, None, None]
        jj = np.arange(bd.js - bd.ng, bd.je + bd.ng + 1, dtype=np.float64)[None, :, None]
        dist = ((zm - zc) / zc)[None, None, :] ** 2 + ((ii - ic) * dx_const / dt_rad) ** 2 + ((jj - jc) * dy_const / dt_rad) ** 2
        out["pt"] += dt_amp * np.maximum(1.0 - np.sqrt(dist), 0.0)
    return out
