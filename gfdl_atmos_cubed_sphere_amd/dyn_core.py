"""Host orchestration of the acoustic substep loop -- the build's counterpart of ``dyn_core``
(model/dyn_core.F90:94-1393), nonhydrostatic branch, non-nested, ``grid_type=4``.

The order of kernel calls and halo updates follows model/dyn_core.F90:313-1286 line by line (cited at
each step).  All fields are device resident (``lib.DeviceArray``); fields that ``d_sw`` /
``update_dz_d`` update in place in the reference are ping-pong pairs here (``cur``/``nxt``), swapped
after the call -- the halo update that follows in the reference fills the new buffer's halo.

Not reproduced (off in every BASELINE config): nesting / regional BCs, ``breed_vortex_inline``,
``do_fast_phys``, ``Ray_fast``.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from .halo import HaloExchanger
from .lib import CP_AIR, GRAV, KAPPA, RDGAS, Context, nh_consts


@dataclass
class DynFlags:
    """The fv_flags_type members the substep reads (defaults: model/fv_arrays.F90:207-906)."""
    n_split: int = 1
    nord: int = 1
    d4_bg: float = 0.16
    d2_bg: float = 0.0
    d2_bg_k1: float = 0.20   # fv_control.F90:1031-1034 when n_sponge == 0 ... reference default 4. -> 0.20
    d2_bg_k2: float = 0.015
    dddmp: float = 0.0
    vtdm4: float = 0.0
    do_vort_damp: bool = False
    d_con: float = 0.0
    delt_max: float = 1.0    # K/s, fv_arrays.F90:667
    hydrostatic: bool = False
    use_cond: bool = False     # thermostruct%use_cond: q_con is transported by d_sw and enters the Riemann solvers' pm2
    moist_kappa: bool = False  # thermostruct%moist_kappa: per-cell cappa in the Riemann solvers (and the remap)
    d_ext: float = 0.02      # external-mode damping (hydrostatic one_grad_p only), fv_arrays.F90:452
    inline_q: bool = False   # tracers advected inside d_sw every substep instead of tracer_2d (sw_core.F90:1020-1043), fv_arrays.F90:474
    fill_dp: bool = False    # mix_dp behind d_sw (dyn_core.F90:820, :2119-2200): thin layers borrow mass from their neighbour; needs ak, bk
    beta: float = 0.0        # > 0: split_p_grad / grad1_p_update (time-off-centred hydrostatic pressure gradient), fv_arrays.F90:403
    convert_ke: bool = False
    ke_bg: float = 0.0
    hord_mt: int = 10
    hord_vt: int = 10
    hord_tm: int = 10
    hord_dp: int = 10
    hord_tr: int = 8
    a_imp: float = 1.0       # > 0.999: SIM1_solver in Riem_Solver3 (BASELINE north_star); reference default 0.75
    m_split: int = 1         # flagstruct%m_split: the sub-steps of RIM_2D (a_imp <= 0.5)
    p_fac: float = 0.05
    use_logp: bool = False
    use_old_omega: bool = True
    n_sponge: int = 1
    is_ideal_case: bool = False
    ptop: float = 300.0
    akap: float = KAPPA
    grav: float = GRAV
    rdgas: float = RDGAS
    cp_air: float = CP_AIR
    fast_tau_w_sec: float = 0.0   # > 1e-5: Rayleigh damping of w inside SIM1 / SIM (nh_utils.F90:356-367, :1363-1371), fv_arrays.F90:702
    rf_fast: bool = False         # RF_fast: Ray_fast at the end of every acoustic substep when tau > 0 (dyn_core.F90:1057-1060)
    tau: float = 0.0              # days: flagstruct%tau -- Ray_fast here, Rayleigh_Friction / _Super in FvDynamics (which merges its tau argument into this)
    rf_cutoff: float = 30.0e2


def level_coefficients(npz: int, fl: DynFlags) -> dict:
    """nord_k, nord_v, nord_w, nord_t, d2_divg, damp_vt, damp_w, damp_t, d_con_k per level, exactly as the
    k loop of model/dyn_core.F90:666-733 computes them."""
    lev = {k: np.zeros(npz, dtype=np.int32) for k in ("nord_k", "nord_v", "nord_w", "nord_t")}
    lev.update({k: np.zeros(npz) for k in ("d2_divg", "damp_vt", "damp_w", "damp_t", "d_con_k")})
    for k in range(1, npz + 1):
        nord_k = fl.nord
        nord_v = min(2, fl.nord)
        d2_divg = min(0.20, fl.d2_bg)
        damp_vt = fl.vtdm4 if fl.do_vort_damp else 0.0
        nord_w, nord_t, damp_w, damp_t, d_con_k = nord_v, nord_v, damp_vt, damp_vt, fl.d_con
        if npz == 1 or fl.n_sponge < 0:
            d2_divg = fl.d2_bg
        else:
            if k == 1:
                nord_k = 0
                d2_divg = max(fl.d2_bg, fl.d2_bg_k1) if fl.is_ideal_case else max(0.01, fl.d2_bg, fl.d2_bg_k1)
                nord_w, damp_w = 0, d2_divg
                if fl.do_vort_damp:
                    nord_v, damp_vt = 0, 0.5 * d2_divg
                d_con_k = 0.0
            elif k == 2 and fl.d2_bg_k2 > 0.01:
                nord_k = 0
                d2_divg = max(fl.d2_bg, fl.d2_bg_k2)
                nord_w, damp_w = 0, d2_divg
                if fl.do_vort_damp:
                    nord_v, damp_vt = 0, 0.5 * d2_divg
                d_con_k = 0.0
            elif k == 3 and fl.d2_bg_k2 > 0.05:
                nord_k = 0
                d2_divg = max(fl.d2_bg, 0.2 * fl.d2_bg_k2)
                nord_w, damp_w = 0, d2_divg
                d_con_k = 0.0
        i = k - 1
        lev["nord_k"][i], lev["nord_v"][i], lev["nord_w"][i], lev["nord_t"][i] = nord_k, nord_v, nord_w, nord_t
        lev["d2_divg"][i], lev["damp_vt"][i], lev["damp_w"][i], lev["damp_t"][i] = d2_divg, damp_vt, damp_w, damp_t
        lev["d_con_k"][i] = d_con_k
    return lev


class DynCore:
    """Device-resident state + work arrays of one rank and the substep loop."""

    PROGNOSTIC = (("u", "U"), ("v", "V"), ("w", "A"), ("delp", "A"), ("pt", "A"))

    def __init__(self, ctx: Context, flags: DynFlags, dp_ref, px: int = 1, py: int = 1, rank: int = 0, world: int = 1,
                 halo=None, pfull=None, ks: int = 0, akbk=None):
        """pfull (npz; fv_dynamics.F90:254-262) and ks (the levels of pure pressure) are read by fast_tau_w_sec / RF_fast only;
        akbk = (ak, bk) by fill_dp only (dyn_core's own arguments, dyn_core.F90:94-98; FvDynamics hands its own over)"""
        self.ctx, self.fl = ctx, flags
        if akbk is not None:
            ctx.set_ak_bk(np.asarray(akbk[0], dtype=np.float64), np.asarray(akbk[1], dtype=np.float64))
        elif flags.fill_dp:
            raise ValueError("fill_dp (mix_dp) needs ak, bk: DynCore(..., akbk=(ak, bk))")
        self.pfull, self.ks, self.dp_ref = (None if pfull is None else np.asarray(pfull, dtype=np.float64)), int(ks), np.asarray(dp_ref, dtype=np.float64)
        self._rfw_ready = self._rff_ready = False
        if (flags.fast_tau_w_sec > 1.0e-5 or (flags.rf_fast and flags.tau > 0.0)) and pfull is None:
            raise ValueError("fast_tau_w_sec / RF_fast need pfull (and ks)")
        self.npz = ctx.npz
        # halo: an object with HaloExchanger's interface (cubed_dyn.CubeHaloAdapter for the six faces of the sphere)
        self.halo = halo if halo is not None else HaloExchanger(ctx, px, py, rank, world)
        npz = self.npz
        z = ctx.zeros
        d = self.d = {}
        for n, kind in self.PROGNOSTIC:
            d[n] = z(kind, npz)
            d[n + "_nxt"] = z(kind, npz)
        d["delz"] = z("CC", npz)
        d["phis"] = z("A")
        d["zs"] = z("A")
        # dyn_core work arrays (dyn_core.F90:256-283) and fv_atmos_type auxiliaries
        for n, kind, nk in (("delpc", "A", npz), ("ptc", "A", npz), ("uc", "V", npz), ("vc", "U", npz),
                            ("ua", "A", npz), ("va", "A", npz), ("omga", "A", npz), ("ut", "A", npz),
                            ("vt", "A", npz), ("divgd", "B", npz), ("gz", "A", npz + 1), ("pkc", "A", npz + 1),
                            ("zh", "A", npz + 1), ("zh_nxt", "A", npz + 1), ("pk3", "A", npz + 1),
                            ("crx", "CX", npz), ("xfx", "CX", npz), ("cry", "CY", npz), ("yfx", "CY", npz),
                            ("mfx", "FX", npz), ("mfy", "FY", npz), ("cx", "CX", npz), ("cy", "CY", npz),
                            ("heat_s", "CC", npz), ("diss_e", "CC", npz), ("pk", "CC", npz + 1),
                            ("ws3", "A", None), ("ws", "CC", None)):
            d[n] = z(kind, nk)
        b = ctx.bd
        d["pe"] = ctx.from_host(np.zeros((b.nx + 2, npz + 1, b.ny + 2), order="F"))
        d["peln"] = ctx.from_host(np.zeros((b.nx, npz + 1, b.ny), order="F"))
        if flags.use_cond:
            d["q_con"], d["q_con_nxt"] = z("A", npz), z("A", npz)
        if flags.moist_kappa:
            d["cappa"] = z("A", npz)
        if flags.beta < 0.0 and (flags.hydrostatic or flags.beta >= -0.1):
            raise ValueError("beta < 0: only beta < -0.1 in the nonhydrostatic loop selects anything (one_grad_p, dyn_core.F90:1029)")
        if flags.beta > 1.0e-9:   # dyn_core.F90:278-283: allocated and zeroed once, kept between calls
            d["du"], d["dv"] = z("U", npz), z("V", npz)
        self.lev = level_coefficients(npz, flags)
        ctx.dsw_levels(self.lev)
        ctx.set_dp_ref(dp_ref)
        self.cn = nh_consts(flags.ptop, p_fac=flags.p_fac, a_imp=flags.a_imp, akap=flags.akap, grav=flags.grav,
                            rdgas=flags.rdgas, cp_air=flags.cp_air, m_split=getattr(flags, "m_split", 1))

    # -- the damping profiles the reference evaluates on the first call and keeps ------------------------
    def _diss_est_begin(self) -> bool:
        """flagstruct%do_diss_est (the SKEB dissipation estimate; a member of the gridstruct the context uploaded): d_sw returns diss_e of
        every level, dyn_core sums it into diss_est over the acoustic substeps (dyn_core.F90:805-811); diss_est is zeroed on the first
        call only (init_step, :285) and keeps accumulating over the calls after it, as the reference's does."""
        if not getattr(self.ctx.grid, "do_diss_est", False):
            return False
        if "diss_est" not in self.d:
            self.d["diss_est"] = self.ctx.zeros("A", self.npz)
        return True

    def fast_tau_w_profile(self, dt_c: float):
        """rff(1:k_rf) of nh_utils.F90:356-367: Riem_Solver_c's first call, with ITS dt (half the acoustic step)"""
        fl, rff = self.fl, []
        for pf in self.pfull:
            if pf > fl.rf_cutoff:
                break
            rff_temp = dt_c / fl.fast_tau_w_sec * np.sin(0.5 * np.pi * np.log(fl.rf_cutoff / pf) / np.log(fl.rf_cutoff / fl.ptop)) ** 2
            rff.append(1.0 / (1.0 + rff_temp))
        return np.array(rff)

    def ray_fast_profile(self, dt: float):
        """(kmax, k_rf, dm, rf) of Ray_fast's first call (dyn_core.F90:2519-2545)"""
        fl, npz = self.fl, self.npz
        tau0 = fl.tau * 86400.0
        rf, kmax = np.ones(npz), 1
        for k, pf in enumerate(self.pfull):
            if pf < fl.rf_cutoff:
                rffk = dt / tau0 * np.sin(0.5 * np.pi * np.log(fl.rf_cutoff / pf) / np.log(fl.rf_cutoff / fl.ptop)) ** 2
                kmax = k + 1
                rf[k] = 1.0 / (1.0 + rffk)
            else:
                break
        dm, k_rf = 0.0, 0
        for k in range(self.ks):
            if self.pfull[k] < fl.rf_cutoff + min(100.0, 10.0 * fl.ptop):
                dm = dm + float(self.dp_ref[k])
                k_rf = k + 1
            else:
                break
        return kmax, k_rf, dm, rf

    def _ray_fast(self, dt: float):
        """dyn_core.F90:1057-1060"""
        fl = self.fl
        if not (fl.rf_fast and fl.tau > 0.0):
            return
        if not self._rff_ready:
            kmax, k_rf, dm, rf = self.ray_fast_profile(abs(dt))
            self.ctx.set_ray_fast(kmax, k_rf, dm if k_rf > 0 else 1.0, rf[:kmax], self.dp_ref)
            self._rff_ready = True
        self.ctx.ray_fast(self.d["u"], self.d["v"], None if fl.hydrostatic else self.d["w"], fl.hydrostatic)

    # -- state I/O ------------------------------------------------------------------------------------
    def set_state(self, u, v, w, delp, pt, delz, phis):
        d = self.d
        for n, a in (("u", u), ("v", v), ("w", w), ("delp", delp), ("pt", pt), ("delz", delz), ("phis", phis)):
            d[n].upload(a)
        if isinstance(phis, (list, tuple)):   # six faces
            d["zs"].upload([np.asfortranarray(p * (1.0 / self.fl.grav)) for p in phis])
        else:
            d["zs"].upload(np.asfortranarray(phis * (1.0 / self.fl.grav)))  # dyn_core.F90:246-251

    def get_state(self):
        return {n: self.d[n].download() for n in ("u", "v", "w", "delp", "pt", "delz", "zh", "mfx", "mfy", "cx", "cy")}

    def _swap(self, name):
        self.d[name], self.d[name + "_nxt"] = self.d[name + "_nxt"], self.d[name]

    # -- inline_q (dyn_core.F90:340 / :573 / :768, sw_core.F90:1020-1043) ------------------------------------------------
    def _inline_q(self):
        """the tracers ride inside d_sw: needs the tracer pair of the model step (FvDynamics puts q, q_nxt into self.d)"""
        return self.fl.inline_q and "q" in self.d

    def _inline_q_fluxes(self):
        """d_sw writes this substep's delp fluxes into zeroed arrays of their own (the tracers' mass fluxes)"""
        d = self.d
        for n, kind in (("fx_s", "FX"), ("fy_s", "FY")):
            if n not in d:
                d[n] = self.ctx.zeros(kind, self.npz)
            d[n].zero()
        return d["fx_s"], d["fy_s"]

    def _inline_q_transport(self):
        """after d_sw, before the swap of delp: q -> q_nxt with d_sw's per-substep Courant numbers and fluxes; mfx += fx"""
        d, fl, ctx = self.d, self.fl, self.ctx
        nq = d["q"].shape[3]
        ctx.d_sw_inline_q(nq, fl.hord_tr, int(self.lev["nord_t"][0]), float(self.lev["damp_t"][0]), d["q"], d["q_nxt"], d["delp"],
                          d["delp_nxt"], d["fx_s"], d["fy_s"], d["crx"], d["cry"], d["xfx"], d["yfx"])
        ctx.flux_accum(d["mfx"], d["mfy"], d["fx_s"], d["fy_s"])
        self._swap("q")

    # -- the substep loop, hydrostatic branch (dyn_core.F90:313-1286 with hydrostatic = .true., beta = 0) ----------
    def run_hydrostatic(self, bdt: float):
        fl, d, ctx, halo = self.fl, self.d, self.ctx, self.halo
        n_split = fl.n_split
        dt = bdt / float(n_split)
        dt2 = 0.5 * dt
        ptk = fl.ptop ** fl.akap
        for a in ("mfx", "mfy", "cx", "cy"):
            d[a].zero()
        for n, kind, nk in (("pkz", "CC", self.npz), ("divg2", "A", None)):
            if n not in d:
                d[n] = ctx.zeros(kind, nk)
        heating = fl.d_con > 1.0e-5
        if heating:
            if "heat_source" not in d:
                d["heat_source"] = ctx.zeros("A", self.npz)
            d["heat_source"].zero()
        diss = self._diss_est_begin()
        par = dict(dt=dt, hord_tr=fl.hord_tr, hord_mt=fl.hord_mt, hord_vt=fl.hord_vt, hord_tm=fl.hord_tm,
                   hord_dp=fl.hord_dp, dddmp=fl.dddmp, d4_bg=fl.d4_bg, kgb=fl.ke_bg, hydrostatic=1, use_cond=0)
        halo.update([(d["delp"], "A"), (d["pt"], "A")])
        halo.update([(d["u"], "U"), (d["v"], "V")])
        for it in range(1, n_split + 1):
            remap_step = it == n_split
            ctx.c_sw(d["delpc"], d["delp"], d["ptc"], d["pt"], d["u"], d["v"], None, d["uc"], d["vc"], d["ua"],
                     d["va"], None, d["ut"], d["vt"], d["divgd"], fl.nord, dt2, True)         # :439-447
            if fl.nord > 0:
                halo.update([(d["divgd"], "B")])
            ctx.geopk(fl.ptop, fl.akap, fl.cp_air, d["pe"], d["peln"], d["delpc"], d["pkc"], d["gz"], d["phis"], d["ptc"],
                      d["pkz"], True)                                         # :480-482 (CG)
            ctx.p_grad_c(dt2, d["delpc"], d["pkc"], d["gz"], d["uc"], d["vc"], True)           # :562
            inline = self._inline_q()
            if inline:
                halo.update([(d["q"], "A")])                                  # :341 start ... :573 complete (pack 10)
            mfx, mfy = self._inline_q_fluxes() if inline else (d["mfx"], d["mfy"])
            dsw_args = (par, d["vt"], d["delp"], d["pt"], d["u"], d["v"], None, d["uc"], d["vc"], d["ua"], d["va"],
                        d["divgd"], mfx, mfy, d["cx"], d["cy"], d["crx"], d["cry"], d["xfx"], d["yfx"], None,
                        d["delp_nxt"], d["pt_nxt"], d["u_nxt"], d["v_nxt"], None, None, d["heat_s"] if heating else None,
                        d["diss_e"] if diss else None)   # (heat_s is read only when d_con > 1e-5, diss_e with do_diss_est: :798-812)
            if halo.overlaps:      # :565 / :578 (pack 9) around the interior of d_sw (:762), as in the nonhydrostatic loop
                pending = halo.start([(d["uc"], "V"), (d["vc"], "U")], defer=True)
                ctx.d_sw(*dsw_args, phase="interior")
                halo.post(pending)
                halo.finish(pending)
                ctx.d_sw(*dsw_args, phase="rest")
            else:
                halo.update([(d["uc"], "V"), (d["vc"], "U")])
                ctx.d_sw(*dsw_args)                                           # :762
            if heating:
                ctx.heat_source_accum(d["heat_source"], d["heat_s"])
            if diss:
                ctx.heat_source_accum(d["diss_est"], d["diss_e"])             # :805-811
            if inline:
                self._inline_q_transport()
            # external-mode damping field from the delp BEFORE d_sw (:745-747) and d_sw's divergence output (:791-848)
            ctx.divg2_ext(fl.d_ext, d["delp"], d["vt"], d["divg2"])
            for n in ("delp", "pt", "u", "v"):
                self._swap(n)
            if fl.fill_dp:
                ctx.mix_dp(True, None, d["delp"], d["pt"])                    # :820
            halo.update([(d["delp"], "A"), (d["pt"], "A")])                   # :823-824 / :851
            ctx.geopk(fl.ptop, fl.akap, fl.cp_air, d["pe"], d["peln"], d["delp"], d["pkc"], d["gz"], d["phis"], d["pt"],
                      d["pkz"], False)                                        # :905-907
            if remap_step:
                ctx.copy_a_to_cc(d["pkc"], d["pk"])                           # pk = pkc, :1001-1010
            if fl.beta > 0.0:   # :1018-1019, beta_d = 0 in the first substep (:398-406)
                ctx.grad1_p_update(d["divg2"] if fl.d_ext > 0.0 else None, d["u"], d["v"], d["pkc"], d["gz"], dt, ptk,
                                   0.0 if it == 1 else fl.beta, d["du"], d["dv"])
            else:
                ctx.one_grad_p(d["u"], d["v"], d["pkc"], d["gz"], d["divg2"] if fl.d_ext > 0.0 else None, dt, ptk)  # :1021
            self._ray_fast(dt)                                                # :1057-1060
            if it != n_split:
                halo.update([(d["u"], "U"), (d["v"], "V")])
            elif hasattr(halo, "sync_edges"):
                halo.sync_edges(d["u"], d["v"])                               # mpp_get_boundary, :1151-1163 (cubed sphere)
        n_con = self.n_con()
        if n_con != 0 and heating:
            halo.update([(d["heat_source"], "A")])
            ctx.del2_cubed(d["heat_source"], 0.20 * ctx.grid.da_min, min(3, fl.nord + 1))
            ctx.apply_heat_source(n_con, True, bdt, fl.delt_max, fl.cp_air, fl.cp_air - fl.rdgas, fl.rdgas, fl.grav,
                                  d["pt"], d["heat_source"], d["delp"], None, d["pkz"])

    # -- the substep loop (dyn_core.F90:313-1286) ------------------------------------------------------
    def run(self, bdt: float, end_step: bool = True):
        if self.fl.hydrostatic:
            return self.run_hydrostatic(bdt)
        fl, d, ctx, halo = self.fl, self.d, self.ctx, self.halo
        n_split = fl.n_split
        dt = bdt / float(n_split)
        dt2 = 0.5 * dt
        rdt = 1.0 / dt
        ptk = fl.ptop ** fl.akap             # dyn_core.F90:222
        peln1 = np.log(fl.ptop)
        for a in ("mfx", "mfy", "cx", "cy"):  # :289-292 empty the flux capacitors
            d[a].zero()
        if fl.beta < -0.1 and fl.d_ext > 0.0 and "divg2" not in d:
            d["divg2"] = ctx.zeros("A", None)
        heating = fl.d_con > 1.0e-5
        if heating:                            # :294
            if "heat_source" not in d:
                d["heat_source"] = ctx.zeros("A", self.npz)
            if "pkz" not in d:
                d["pkz"] = ctx.zeros("CC", self.npz)
            d["heat_source"].zero()
        diss = self._diss_est_begin()
        par = dict(dt=dt, hord_tr=fl.hord_tr, hord_mt=fl.hord_mt, hord_vt=fl.hord_vt, hord_tm=fl.hord_tm,
                   hord_dp=fl.hord_dp, dddmp=fl.dddmp, d4_bg=fl.d4_bg, kgb=fl.ke_bg, hydrostatic=0,
                   use_cond=int(fl.use_cond))
        cond = lambda: ctx.set_condensate(d["q_con"] if fl.use_cond else None, d["cappa"] if fl.moist_kappa else None)
        # fv_dynamics.F90:467-470: halo of delp, pt (pack 1) and u, v (pack 8) before the first substep
        halo.update([(d["delp"], "A"), (d["pt"], "A")])
        halo.update([(d["u"], "U"), (d["v"], "V")])
        for it in range(1, n_split + 1):
            remap_step = it == n_split
            halo.update([(d["w"], "A")])                                      # :350 / :432 (pack 7)
            if it == 1:                                                       # :353-389
                ctx.zh_from_delz(d["zs"], d["delz"], d["zh"])
                halo.update([(d["zh"], "A")])                                 # gz halo (pack 5), then zh = gz (:491-499)
            ctx.c_sw(d["delpc"], d["delp"], d["ptc"], d["pt"], d["u"], d["v"], d["w"], d["uc"], d["vc"], d["ua"],
                     d["va"], d["omga"], d["ut"], d["vt"], d["divgd"], fl.nord, dt2, False)   # :439-447
            if fl.nord > 0:
                halo.update([(d["divgd"], "B")])                              # :451 / :577 (pack 3, CORNER)
            ctx.update_dz_c(dt2, d["zs"], d["ut"], d["vt"], d["zh"], d["gz"], d["ws3"])       # :514-527
            cond()
            if fl.fast_tau_w_sec > 1.0e-5 and not self._rfw_ready:             # nh_utils.F90:356-367, once
                ctx.set_fast_tau_w(self.fast_tau_w_profile(dt2))
                self._rfw_ready = True
            ctx.riem_solver_c(dt2, self.cn, d["phis"], d["omga"], d["ptc"], d["delpc"], d["gz"], d["pkc"], d["ws3"])  # :531
            ctx.p_grad_c(dt2, d["delpc"], d["pkc"], d["gz"], d["uc"], d["vc"], False)         # :562
            # :565 / :578 (pack 9, CGRID_NE) overlapped with the interior of d_sw (:762): start ... complete
            inline = self._inline_q()
            if inline:
                halo.update([(d["q"], "A")])                                  # :341 start ... :573 complete (pack 10)
            mfx, mfy = self._inline_q_fluxes() if inline else (d["mfx"], d["mfy"])
            dsw_args = (par, d["vt"], d["delp"], d["pt"], d["u"], d["v"], d["w"], d["uc"], d["vc"], d["ua"], d["va"],
                        d["divgd"], mfx, mfy, d["cx"], d["cy"], d["crx"], d["cry"], d["xfx"], d["yfx"],
                        d["q_con"] if fl.use_cond else None,
                        d["delp_nxt"], d["pt_nxt"], d["u_nxt"], d["v_nxt"], d["w_nxt"],
                        d["q_con_nxt"] if fl.use_cond else None, d["heat_s"] if heating else None,
                        d["diss_e"] if diss else None)   # (heat_s is read only when d_con > 1e-5, diss_e with do_diss_est: :798-812)
            if halo.overlaps:
                pending = halo.start([(d["uc"], "V"), (d["vc"], "U")], defer=True)
                ctx.d_sw(*dsw_args, phase="interior")
                halo.post(pending)
                halo.finish(pending)
                ctx.d_sw(*dsw_args, phase="rest")
            else:
                halo.update([(d["uc"], "V"), (d["vc"], "U")])
                ctx.d_sw(*dsw_args)
            if heating:
                ctx.heat_source_accum(d["heat_source"], d["heat_s"])          # :798-803
            if diss:
                ctx.heat_source_accum(d["diss_est"], d["diss_e"])             # :805-811: diss_est(i,j,k) += diss_e(i,j)
            if inline:
                self._inline_q_transport()
            if fl.beta < -0.1 and fl.d_ext > 0.0:   # :745-747, :791-848: the external-mode damping field of one_grad_p (:1030)
                ctx.divg2_ext(fl.d_ext, d["delp"], d["vt"], d["divg2"])
            for n in ("delp", "pt", "u", "v", "w") + (("q_con",) if fl.use_cond else ()):
                self._swap(n)
            if fl.fill_dp:
                ctx.mix_dp(False, d["w"], d["delp"], d["pt"])                 # :820
            # :823-825 start / :851-852 complete (packs 1, 11).  Several ranks: the messages stay in flight while update_dz_d
            # and Riem_Solver3 run -- neither reads a halo of delp, pt, q_con (column kernels over the compute domain; the
            # transport of zh reads the Courant numbers and area fluxes) -- and are completed in front of pk3_halo, the
            # first reader of delp's halo
            grp1 = [(d["delp"], "A"), (d["pt"], "A")] + ([(d["q_con"], "A")] if fl.use_cond else [])
            lag = bool(getattr(halo, "overlaps_groups", False))
            pend1 = halo.start(grp1, defer=True) if lag else halo.update(grp1)
            if fl.use_cond:
                cond()
            ctx.update_dz_d(fl.hord_tm, d["zs"], d["zh"], d["zh_nxt"], d["crx"], d["cry"], d["xfx"], d["yfx"],
                            d["ws"], rdt)                                     # :911
            self._swap("zh")
            if lag:
                halo.post(pend1)
            ctx.riem_solver3(dt, self.cn, d["zs"], d["w"], d["delz"], d["pt"], d["delp"], d["zh"], d["pe"], d["pkc"],
                             d["pk3"], d["pk"], d["peln"], d["ws"], fl.use_logp, remap_step, fl.beta < -0.1)   # :932, fp_out :939
            if lag:
                halo.finish(pend1)
            # :944-950 (packs 4, 5) start ... complete around pe_halo / pk3_halo, which read delp only
            grp2 = [(d["zh"], "A"), (d["pkc"], "A")]
            pend2 = halo.start(grp2, defer=True) if lag else halo.update(grp2)
            if remap_step:
                ctx.pe_halo(fl.ptop, d["pe"], d["delp"])                      # :952-953
            ctx.pk3_halo(fl.ptop, fl.akap, d["pk3"], d["delp"], fl.use_logp)  # :955-959
            if lag:
                halo.post(pend2)
                halo.finish(pend2)
            # :982-989 gz = zh*grav is fused into nh_p_grad (gz_scale)
            if fl.beta > 0.0:   # :1027-1028, beta_d = 0 in the first substep (:398-406)
                ctx.split_p_grad(d["u"], d["v"], d["pkc"], d["zh"], d["delp"], d["pk3"], 0.0 if it == 1 else fl.beta, dt,
                                 peln1 if fl.use_logp else ptk, d["du"], d["dv"], gz_scale=fl.grav)
            elif fl.beta < -0.1:   # :1029-1030: pkc is the full pressure (fp_out), the layer weights a2b_ord4 of delp
                ctx.one_grad_p_nh(d["u"], d["v"], d["pkc"], d["zh"], d["divg2"] if fl.d_ext > 0.0 else None, d["delp"], dt,
                                  fl.ptop, gz_scale=fl.grav)
            else:
                ctx.nh_p_grad(d["u"], d["v"], d["pkc"], d["zh"], d["delp"], d["pk3"], dt,
                              peln1 if fl.use_logp else ptk, gz_scale=fl.grav)    # :1032
            self._ray_fast(dt)                                                # :1057-1060
            if it != n_split:
                halo.update([(d["u"], "U"), (d["v"], "V")])                   # :1168-1169 (pack 8)
            else:
                if hasattr(halo, "sync_edges"):
                    halo.sync_edges(d["u"], d["v"])                           # mpp_get_boundary, :1151-1163 (cubed sphere)
                if fl.use_old_omega and end_step:                             # last_step = it==n_split .and. end_step (:427)
                    # :1182-1191: omga = (pe - pem)*rdt; pem = p of the delp this substep started from (:409-421), which
                    # the ping-pong left in delp_nxt
                    ctx.omga_update(rdt, fl.ptop, d["pe"], d["delp_nxt"], d["omga"])
                    if ctx.grid.grid_type < 3 and "en1" in ctx.grid.m:         # :1195 adv_pe (en1 / en2 exist on the cubed sphere only)
                        ctx.adv_pe(fl.ptop, d["ua"], d["va"], d["delp_nxt"], d["omga"])
        # ---- dissipative heating (:296-308, :1300-1355) ----
        n_con = self.n_con()
        if n_con != 0 and heating:
            halo.update([(d["heat_source"], "A")])                            # del2_cubed's mpp_update_domains, :2399
            ctx.del2_cubed(d["heat_source"], 0.20 * ctx.grid.da_min, min(3, fl.nord + 1))    # :1301-1303
            ctx.apply_heat_source(n_con, False, bdt, fl.delt_max, fl.cp_air, fl.cp_air - fl.rdgas, fl.rdgas, fl.grav,
                                  d["pt"], d["heat_source"], d["delp"], d["delz"], d["pkz"])

    def n_con(self) -> int:
        """number of levels that receive the dissipative heating (dyn_core.F90:296-308)"""
        fl = self.fl
        if fl.convert_ke or (fl.do_vort_damp and fl.vtdm4 > 1.0e-4):
            return self.npz
        if fl.d2_bg_k1 < 1.0e-3:
            return 0
        return 1 if fl.d2_bg_k2 < 1.0e-3 else 2
