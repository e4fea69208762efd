"""Host orchestration of one atmospheric time step -- the build's counterpart of the k_split loop of
``fv_dynamics`` (model/fv_dynamics.F90:460-665): for each remapping cycle the acoustic substeps (``dyn_core``),
the sub-cycled tracer transport (``tracer_2d``) and the vertical remap (``Lagrangian_to_Eulerian``).

``step`` expects the state in the form ``dyn_core`` works on (pt = theta_v, delz < 0); ``step_from_temperature`` is a whole
``fv_dynamics`` call on a Cartesian domain or on the six faces of the sphere: compute_total_energy when consv_te > 0 (:345), T ->
theta_v / theta_m (fv_dynamics.F90:296-399, with moist_cv under use_cond / moist_kappa), the Rayleigh damping when tau > 0
(:362-371: Rayleigh_Super on the cubed sphere and in ideal cases, Rayleigh_Friction otherwise), the k_split loop, the energy
fixer and theta_v -> T in the last remap, cubed_to_latlon (:911).  The angular-momentum fixer (consv_am), nudging and the
diagnostics (:669-803) are not built.
"""
from __future__ import annotations

import numpy as np

from .dyn_core import DynCore, DynFlags
from .lib import CP_AIR, Context
from .tracer2d import tracer_2d


CONSV_MIN = 0.001   # fv_mapz.F90:45: below it no correction applies


class FvDynamics:
    def __init__(self, ctx: Context, flags: DynFlags, ak, bk, nq: int = 0, k_split: int = 1, kord_tm: int = -8,
                 kord_mt: int = 8, kord_wz: int = 8, kord_tr: int = 8, q_split: int = 0, nord_tr: int = 0, trdm2: float = 0.0, adiabatic: bool = True,
                 px: int = 1, py: int = 1, rank: int = 0, world: int = 1, dist=None, tau: float = 0.0,
                 rf_cutoff: float = 30.0e2, c2l_ord: int = 4, moist: dict | None = None, fill: bool = False, halo=None,
                 consv_te: float = 0.0, moist_phys: bool = False, radius: float = 6.3712e6, fill2d: tuple = (),
                 remap_te: bool = False, consv_am: dict | None = None):
        ak, bk = np.asarray(ak, dtype=np.float64), np.asarray(bk, dtype=np.float64)
        dp_ref = (ak[1:] - ak[:-1]) + (bk[1:] - bk[:-1]) * 1.0e5          # dyn_core.F90:241-244
        # flagstruct%tau / %rf_cutoff are ONE pair in the reference: Rayleigh_Friction / _Super here (fv_dynamics.F90:362-376) and Ray_fast /
        # fast_tau_w_sec inside dyn_core (dyn_core.F90:1057-1060) read the same numbers.  The argument and the flags' member are merged
        # (either may carry the value; two different values are an error), so that tau > 0 with rf_fast damps through Ray_fast
        # instead of silently not at all.
        import dataclasses
        if tau != 0.0 and flags.tau != 0.0 and tau != flags.tau:
            raise ValueError(f"FvDynamics: tau = {tau} but flags.tau = {flags.tau} (flagstruct%tau is one number)")
        tau = tau if tau != 0.0 else flags.tau
        rf_default = DynFlags.__dataclass_fields__["rf_cutoff"].default
        if rf_cutoff != rf_default and flags.rf_cutoff != rf_default and rf_cutoff != flags.rf_cutoff:
            raise ValueError(f"FvDynamics: rf_cutoff = {rf_cutoff} but flags.rf_cutoff = {flags.rf_cutoff}")
        rf_cutoff = rf_cutoff if rf_cutoff != rf_default else flags.rf_cutoff
        flags = dataclasses.replace(flags, tau=tau, rf_cutoff=rf_cutoff)
        self.ctx, self.fl, self.nq, self.k_split, self.q_split, self.dist = ctx, flags, nq, k_split, q_split, dist
        self.nord_tr, self.trdm2 = nord_tr, trdm2
        # fill2D (fv_dynamics.F90:542-556, FILL2D builds): the 0-based tracer indices of liq_wat, rainwat, ice_wat, snowwat, graupel
        # that get the diffusive filling after tracer_2d when hord_tr < 8 and moist_phys
        self.fill2d = tuple(int(i) for i in fill2d)
        # flagstruct%remap_te (fv_arrays.F90:399): the remap carries total energy in the place of T_v / theta_v
        self.remap_te = bool(remap_te)
        self.tau, self.rf_cutoff, self.c2l_ord = tau, rf_cutoff, c2l_ord
        self.ak, self.bk = ak, bk
        self._rf = None                                                    # (rf, pm, kmax): set on first use, as RF_initialized
        # total-energy conservation (fv_dynamics.F90:345-355, fv_mapz.F90:643-772): fraction of the energy lost in a step that
        # the last remap returns as heat (> consv_min), or a prescribed flux in W/m**2 (< -consv_min)
        self.consv_te, self.moist_phys, self.radius = consv_te, moist_phys, radius
        self.e_flux = 0.0
        # flagstruct%consv_am (fv_dynamics.F90:358-361, :747-800): dict(coslat = cos(agrid(:,:,2)) (A), l2c_u (U), l2c_v (V), zxg (compute
        # domain; idiag%zxg, the mountain-torque term), omega) -- host arrays of the grid (lists of six on the sphere), uploaded on first use
        self.consv_am = consv_am
        self.last_u00 = 0.0
        # moist thermodynamics (flags.use_cond / flags.moist_kappa): nwat and the water-species indices for moist_cv,
        # cv_vap, c_liq, c_ice (lib.Context.set_moist)
        self.moist = None
        if flags.use_cond or flags.moist_kappa:
            self.moist = dict(moist or {}, moist_kappa=int(flags.moist_kappa), use_cond=int(flags.use_cond), sphum=1)
        ph = np.asarray(ak, dtype=np.float64) + np.asarray(bk, dtype=np.float64) * 1.0e5     # fv_dynamics.F90:254-262 (p_ref = 1e5)
        pfull = (ph[1:] - ph[:-1]) / np.log(ph[1:] / ph[:-1])
        ks = int(np.argmax(np.asarray(bk) != 0.0)) - 1 if np.any(np.asarray(bk) != 0.0) else len(pfull)   # the last interface of pure pressure
        self.dc = DynCore(ctx, flags, dp_ref, px, py, rank, world, halo=halo, pfull=pfull, ks=max(ks, 0), akbk=(ak, bk))
        ctx.set_ak_bk(ak, bk)
        npz = ctx.npz
        d = self.dc.d
        d["ps"], d["pkz"] = ctx.zeros("A"), ctx.zeros("CC", npz)
        d["dp1"], d["dp1_nxt"] = ctx.zeros("A", npz), ctx.zeros("A", npz)
        if nq:
            shp = ctx.bd.shape("A", npz) + (nq,)
            d["q"], d["q_nxt"] = ctx.from_host(np.zeros(shp, order="F")), ctx.from_host(np.zeros(shp, order="F"))
        self.remap_par = dict(hydrostatic=int(flags.hydrostatic), adiabatic=int(adiabatic), nq=nq, kord_mt=kord_mt, kord_wz=kord_wz,
                              kord_tm=kord_tm, sphum=1 if nq else 0, akap=flags.akap, ptop=flags.ptop,
                              rdgas=flags.rdgas, grav=flags.grav, cv_air=flags.cp_air - flags.rdgas, r_vir=0.6077,
                              cp=flags.cp_air, t_min=184.0, kord_tr=[kord_tr] * nq, fill=int(fill))

    def set_tracers(self, q: np.ndarray):
        self.dc.d["q"].upload(q)

    def step_from_temperature(self, bdt: float):
        """A whole fv_dynamics call for the adiabatic core: pt holds T (T_v) on entry and on return.
        fv_dynamics.F90:284-399 (T -> theta_v), the k_split loop, and the theta_v -> T conversion that the last remap
        does (fv_mapz.F90:793-821, last_step)."""
        d, ctx, fl = self.dc.d, self.ctx, self.fl
        qv = d["q"] if (self.nq and self.remap_par["sphum"] > 0 and not self.remap_par["adiabatic"]) else None   # first tracer = sphum
        zvir = self.remap_par["r_vir"] if qv else 0.0
        if self.moist:
            ctx.set_moist(self.moist, d.get("q_con"), d.get("cappa"))
        if self.consv_te > CONSV_MIN:                                      # :345-355 (te_2d -> te0_2d of the last remap)
            self.total_energy_before()
        conv = lambda mode: ctx.pt_to_theta_v(mode, zvir, fl.akap, fl.rdgas, fl.grav, d["pt"], d["delp"],
                                              None if fl.hydrostatic else d["delz"], qv, d["pkz"])
        if self.consv_am:                                                  # :358-361: teq, ps2 of the state the step starts from
            self._aam("teq", "ps2")
        if self.tau > 0.0 and not self.fl.rf_fast:                         # :362-376 (RF_fast: Ray_fast inside dyn_core instead) (grid_type = 4: Rayleigh_Friction)
            if not fl.hydrostatic:
                conv(-1)                                                   # pkz from the T, delz before the friction (:323-326)
            self.rayleigh_friction(bdt)
            conv(1)                                                        # :389-397 with that pkz
        else:
            conv(int(fl.hydrostatic))
        self.step(bdt, last_cycle_is_last_step=True)
        if self.consv_am:                                                  # :747-800
            self._consv_am(bdt)
        self.cubed_to_latlon()                                             # :911

    # -- consv_am -------------------------------------------------------------------------------------------------
    def _aam(self, aam_name: str, ps_name: str):
        """compute_aam (fv_dynamics.F90:1266-1314): cubed_to_latlon(ord 2), then aam, m_fac, ps of every column"""
        d, ctx, fl, ca = self.dc.d, self.ctx, self.fl, self.consv_am
        if "coslat" not in d:
            d["coslat"], d["l2c_u"], d["l2c_v"] = ctx.from_host(ca["coslat"]), ctx.from_host(ca["l2c_u"]), ctx.from_host(ca["l2c_v"])
            d["m_fac"] = ctx.zeros("CC")
        for n, kind in ((aam_name, "CC"), (ps_name, "A")):
            if n not in d:
                d[n] = ctx.zeros(kind)
        ctx.c2l(2, d["u"], d["v"], d["ua"], d["va"])                       # :1287 (mode 1, c2l_ord 2: no halo update)
        ctx.compute_aam(self.radius, ca.get("omega", 7.292e-5), 1.0 / fl.grav, fl.ptop, d["coslat"], d["ua"], d["delp"], d[aam_name],
                        d["m_fac"], d[ps_name])

    def _consv_am(self, bdt: float):
        from .global_sum import g_sum
        d, ctx = self.dc.d, self.ctx
        self._aam("aam2", "ps")
        aslist = lambda x: x if isinstance(x, list) else [x]
        areas = self._areas()
        b = ctx.bd
        cc = lambda a: np.asarray(a)[b.ng:b.ng + b.nx, b.ng:b.ng + b.ny]
        te, teq, ps2, ps = (aslist(d[n].download()) for n in ("aam2", "teq", "ps2", "ps"))
        zxg = aslist(self.consv_am["zxg"])
        dt2 = 0.5 * bdt
        te_2d = [t - q + dt2 * (cc(p2) + cc(p1)) * np.asarray(z) for t, q, p2, p1, z in zip(te, teq, ps2, ps, zxg)]     # :761-767
        amdt = g_sum(te_2d, areas, self.dist)                              # :771
        u00 = -self.radius * amdt / g_sum(aslist(d["m_fac"].download()), areas, self.dist)    # :772
        self.last_u00 = u00
        ctx.consv_am_apply(u00, d["l2c_u"], d["l2c_v"], d["u"], d["v"])    # :784-798

    # -- consv_te -------------------------------------------------------------------------------------------------
    def total_energy_before(self):
        """compute_total_energy at fv_dynamics.F90:345 (pt = T, before the conversion to theta_v): te0_2d of every column"""
        d, ctx, hyd = self.dc.d, self.ctx, self.fl.hydrostatic
        if "te0_2d" not in d:
            for n in ("te0_2d", "te_2d", "zsum1", "zsum0"):
                d[n] = ctx.zeros("CC")
        if self.moist:
            ctx.set_moist(self.moist, d.get("q_con"), d.get("cappa"))
        ctx.compute_total_energy(self.remap_par, self.moist_phys, d["u"], d["v"], None if hyd else d["w"],
                                 None if hyd else d["delz"], d["pt"], d["delp"], d.get("q"), None, d["pe"] if hyd else None,
                                 d["peln"] if hyd else None, d["phis"], d["te0_2d"])
        self._te0_valid = True          # consumed (and cleared) by the energy fixer of this call's last remap

    def _areas(self):
        """area (compute domain) of every context: the weights of g_sum (area_64, fv_mapz.F90:736)"""
        ctxs = getattr(self.ctx, "ctxs", [self.ctx])
        out = []
        for c in ctxs:
            b = c.bd
            out.append(np.asarray(c.grid.m["area"])[b.ng:b.ng + b.nx, b.ng:b.ng + b.ny])
        return out

    def _energy_fixer(self, par: dict, pdt: float):
        """fv_mapz.F90:643-772 after the remap of the last step, then step 9a with dtmp (:793-821)"""
        from .global_sum import g_sum
        d, ctx, hyd = self.dc.d, self.ctx, self.fl.hydrostatic
        only_sums = self.consv_te < 0.0
        if not only_sums and not getattr(self, "_te0_valid", False):
            # te_2d = te0_2d - E(remapped column): without the energy of THIS call's initial state (fv_dynamics.F90:345-355, which
            # step_from_temperature does and a bare step() does not) dtmp would be -E / zsum, applied to pt.  A te0_2d left from an
            # earlier call is as wrong.
            raise RuntimeError("consv_te > 0: total_energy_before() was not called for this step (te0_2d unset or stale); "
                               "use step_from_temperature(), or call total_energy_before() on the temperature state first")
        self._te0_valid = False
        for n in ("te0_2d", "te_2d", "zsum1", "zsum0"):
            if n not in d:
                d[n] = ctx.zeros("CC")
        ctx.energy_fixer_sums(par, only_sums, d["u"], d["v"], None if hyd else d["w"], None if hyd else d["delz"], d["pt"],
                              d["delp"], d.get("q"), d["pe"] if hyd else None, d["peln"] if hyd else None, d["phis"], d["pkz"],
                              d["pk"] if hyd else None, d["te0_2d"], d["te_2d"], d["zsum1"], d["zsum0"] if hyd else None)
        aslist = lambda x: x if isinstance(x, list) else [x]
        areas = self._areas()
        zs = g_sum(aslist(d["zsum0" if hyd else "zsum1"].download()), areas, self.dist)
        if only_sums:                                                      # :745-771: a prescribed flux
            self.e_flux = self.consv_te
            dtmp = self.e_flux * (self.fl.grav * pdt * 4.0 * np.pi * self.radius ** 2) / zs
        else:
            dtmp = self.consv_te * g_sum(aslist(d["te_2d"].download()), areas, self.dist)
            self.e_flux = dtmp / (self.fl.grav * pdt * 4.0 * np.pi * self.radius ** 2)
            dtmp = dtmp / zs
        self.dtmp = dtmp
        ctx.remap_finish(par, dtmp, d["pt"], d["pkz"], d.get("q"))

    def rayleigh_profile(self, dt: float):
        """rf(k), kmax of Rayleigh_Friction (fv_dynamics.F90:1169-1182) with pfull of fv_dynamics.F90:254-262 (p_ref = 1e5)"""
        ak, bk, ptop, akap = self.ak, self.bk, self.fl.ptop, self.fl.akap
        ph = ak + bk * 1.0e5
        pfull = (ph[1:] - ph[:-1]) / np.log(ph[1:] / ph[:-1])
        rf = np.zeros(len(pfull))
        kmax = 0
        for k, pm in enumerate(pfull):
            if pm < self.rf_cutoff:
                rf[k] = dt / (self.tau * 86400.0) * np.sin(0.5 * np.pi * np.log(self.rf_cutoff / pm) / np.log(self.rf_cutoff / ptop)) ** 2
                kmax = k + 1
            else:
                break
        return rf, pfull, kmax

    def rayleigh_friction(self, bdt: float, conserve: bool = True):
        """fv_dynamics.F90:362-371: Rayleigh_Super (:953-1124) on the cubed sphere (grid_type < 4) and in ideal cases, Rayleigh_Friction
        (:1126-1264, two kernels around the halo update of u2f) on the Cartesian domains"""
        d, ctx, fl = self.dc.d, self.ctx, self.fl
        if self._rf is None:
            self._rf = self.rayleigh_profile(abs(bdt))
        rf, pm, kmax = self._rf
        if kmax == 0:
            return
        hyd = fl.hydrostatic
        if ctx.grid.grid_type < 4 or fl.is_ideal_case:
            ideal = fl.is_ideal_case
            if ideal and "u00" not in d:                                       # :997-1014: the winds of the first call are kept
                d["u00"], d["v00"] = ctx.zeros("U", ctx.npz), ctx.zeros("V", ctx.npz)
                d["u00"].copy_from(d["u"])
                d["v00"].copy_from(d["v"])
            ctx.c2l(2, d["u"], d["v"], d["ua"], d["va"])                       # :1040-1042
            ctx.rayleigh_super(kmax, not ideal, hyd, fl.cp_air, fl.rdgas, fl.ptop, pm[:kmax], rf[:kmax], d["ua"], d["va"],
                               d["pt"], d["u"], d["v"], None if hyd else d["w"], d.get("u00") if ideal else None,
                               d.get("v00") if ideal else None)
            return
        if "u2f" not in d:
            d["u2f"] = ctx.zeros("A", ctx.npz)
        ctx.rayleigh_u2f(kmax, hyd, d["u"], d["v"], None if hyd else d["w"], d["ua"], d["va"], d["u2f"])
        self.dc.halo.update([(d["u2f"], "A")])                             # :1207-1209
        ctx.rayleigh_apply(kmax, conserve, hyd, fl.cp_air, fl.rdgas, fl.ptop, pm[:kmax], rf[:kmax], d["u2f"], d["pt"],
                           None if hyd else d["delz"], d["u"], d["v"], None if hyd else d["w"])

    def cubed_to_latlon(self):
        """A-grid winds for the physics (fv_dynamics.F90:911; c2l_ord4 updates the halo of u, v first, :2372-2376)"""
        d = self.dc.d
        if self.c2l_ord == 4:
            self.dc.halo.update([(d["u"], "U"), (d["v"], "V")])
        self.ctx.c2l(self.c2l_ord, d["u"], d["v"], d["ua"], d["va"])

    def _fill2d(self):
        """fill2D (fv_fill.F90:183-258) of the water species named at construction: qt = q delp area, its halo (width 1), the fluxes
        between cells of opposite sign and the update of q"""
        d, ctx = self.dc.d, self.ctx
        npz = ctx.npz
        if "qt" not in d:
            d["qt"] = ctx.zeros("A", npz)
        n3 = int(np.prod(d["delp"].shape))
        for iq in self.fill2d:
            ctx.fill2d_mass(npz, d["q"], d["delp"], d["qt"], q_offset=iq * n3)
            self.dc.halo.update([(d["qt"], "A")])                               # :236
            ctx.fill2d_apply(npz, d["qt"], d["delp"], d["q"], q_offset=iq * n3)

    def step(self, bdt: float, last_cycle_is_last_step: bool = False):
        """One dt_atmos: k_split x (n_split acoustic substeps, tracer transport, vertical remap)."""
        d, ctx = self.dc.d, self.ctx
        mdt = bdt / float(self.k_split)
        if "diss_est" in d:                                                    # do_diss_est: dyn_core zeroes it on init_step = (n_map == 1), :497
            d["diss_est"].zero()
        for n_map in range(1, self.k_split + 1):
            last_step = last_cycle_is_last_step and n_map == self.k_split
            d["dp1"].copy_from(d["delp"])                                      # fv_dynamics.F90:475-481
            if self.fl.use_cond:
                self.dc.halo.update([(d["q_con"], "A")])                       # :464 / :487 (pack 11)
            if self.fl.moist_kappa:
                self.dc.halo.update([(d["cappa"], "A")])                       # :465 / :488 (pack 12)
            self.dc.run(mdt, end_step=(n_map == self.k_split))                 # :493, last_step = (n_map == k_split)
            if self.nq and not self.fl.inline_q:                               # :509-533 (inline_q: the tracers rode inside d_sw)
                q, dp1, _ = tracer_2d(ctx, self.dc.halo, d["q"], d["q_nxt"], d["dp1"], d["dp1_nxt"], d["mfx"], d["mfy"],
                                      d["cx"], d["cy"], d["crx"], d["cry"], self.nq, self.fl.hord_tr, self.q_split,
                                      self.nord_tr, self.trdm2, dist=self.dist)
                if q is not d["q"]:
                    d["q"], d["q_nxt"] = d["q_nxt"], d["q"]
                if dp1 is not d["dp1"]:
                    d["dp1"], d["dp1_nxt"] = d["dp1_nxt"], d["dp1"]
                if self.fill2d and self.fl.hord_tr < 8 and self.moist_phys:     # :542-556
                    self._fill2d()
            fixer = last_step and abs(self.consv_te) > CONSV_MIN                # fv_mapz.F90:645-647, :745
            par = dict(self.remap_par, last_step=2 if fixer else int(last_step))
            hyd = self.fl.hydrostatic
            if self.moist:                                                     # q_con is a ping-pong pair: current buffer
                ctx.set_moist(self.moist, d.get("q_con"), d.get("cappa"))
            if self.remap_te:                                                  # te = dp1, as fv_dynamics.F90:612 hands it over
                ctx.set_remap_te(True, d["phis"], d["dp1"])
            ctx.lagrangian_to_eulerian(par, d["ps"], d["pe"], d["delp"], d["pkz"], d["pk"], d["u"], d["v"],
                                       None if hyd else d["w"], None if hyd else d["delz"], d["pt"], d.get("q"),
                                       d["peln"], d["omga"], None if hyd else d["ws"])   # :607
            if fixer:
                self._energy_fixer(par, bdt)
