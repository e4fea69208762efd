"""BASELINE config 2 run for real: the Jablonowski-Williamson baroclinic wave (test_case 13) on the C96 L79 hydrostatic cubed sphere,
whole sphere on one MI355X, `--days` days of dt_atmos = 1800 s (k_split 2 x n_split 6).  Every 12 h:
global air mass (must stay put), min / max surface pressure (JW06: the wave deepens to ~940-950 hPa around day 9), max |u|, max
|v|.  Prints one JSON object."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=96)
    ap.add_argument("--npz", type=int, default=79)
    ap.add_argument("--days", type=float, default=10.0)
    ap.add_argument("--dt-atmos", type=float, default=1800.0)
    ap.add_argument("--k-split", type=int, default=2)
    ap.add_argument("--n-split", type=int, default=6)
    ap.add_argument("--nq", type=int, default=0, help="advected tracers (BASELINE config 5 carries 33): smooth positive fields; eager steps")
    ap.add_argument("--prod", action="store_true", help="the damping set of a production namelist: nord 3, do_vort_damp, vtdm4 0.06, d_con 1, dddmp 0.5")
    ap.add_argument("--flags", type=str, default="", help="JSON of DynFlags overrides")
    ap.add_argument("--nh", action="store_true", help="nonhydrostatic (BASELINE config 3 at --nx 384 --npz 127 --dt-atmos 225 --n-split 5)")
    a = ap.parse_args()
    import torch
    from gfdl_atmos_cubed_sphere_amd import lib as L
    from gfdl_atmos_cubed_sphere_amd.cubed_dyn import CubeHaloAdapter, MultiContext
    from gfdl_atmos_cubed_sphere_amd.cubed_sphere import CubedSphere
    from gfdl_atmos_cubed_sphere_amd.dyn_core import DynFlags
    from gfdl_atmos_cubed_sphere_amd.fv_dynamics import FvDynamics
    from gfdl_atmos_cubed_sphere_amd.test_cases import jablonowski_williamson, set_eta
    nx, npz, npx = a.nx, a.npz, a.nx + 1
    cs = CubedSphere(npx)
    gs = [cs.gridstruct(t) for t in range(6)]
    bd = gs[0].bd
    ak, bk, _, _ = set_eta(npz)
    hyd = not a.nh
    st = jablonowski_williamson(cs, ak, bk, hydrostatic=hyd)
    cs.topo.update("A", [s_["phis"] for s_ in st])
    prod = dict(do_vort_damp=True, vtdm4=0.06, nord=3, d_con=1.0, dddmp=0.5) if a.prod else {}
    if a.flags:
        prod = dict(prod, **json.loads(a.flags))
    fl = DynFlags(n_split=a.n_split, hydrostatic=hyd, ptop=float(ak[0]), **(dict(d_ext=0.0) if hyd else {}), **prod)
    ng = bd.ng
    c = (slice(ng, ng + nx), slice(ng, ng + nx))
    for s_ in st:
        if hyd:
            pe = ak[0] + np.concatenate([np.zeros(s_["delp"].shape[:2] + (1,)), np.cumsum(s_["delp"], axis=2)], axis=2)[c]
            peln = np.log(pe)
            pkz = (pe[:, :, 1:] ** fl.akap - pe[:, :, :-1] ** fl.akap) / (fl.akap * (peln[:, :, 1:] - peln[:, :, :-1]))
        else:
            pkz = ((-fl.rdgas / fl.grav) * s_["delp"][c] * s_["pt"][c] / s_["delz"]) ** fl.akap
        s_["pt"][c] = s_["pt"][c] / pkz
    streams = [torch.cuda.Stream() for _ in range(6)]
    mctx = MultiContext([L.Context(g, npz, stream=fs.cuda_stream) for g, fs in zip(gs, streams)])
    nq = a.nq
    fv = FvDynamics(mctx, fl, ak, bk, nq=nq, k_split=a.k_split, halo=CubeHaloAdapter(mctx, npx, topo=cs.topo))
    if nq:
        q0 = []
        lev = (np.arange(npz) / max(npz - 1, 1))[None, None, :]
        for t in range(6):
            lon, lat = cs.grids[t]["agrid"][..., 0:1], cs.grids[t]["agrid"][..., 1:2]
            q0.append(np.asfortranarray(np.stack([(1.0 + 0.1 * iq) * (1.0 + 0.5 * np.sin((1 + iq % 3) * lon + 0.3 * iq) * np.cos(lat) ** 2
                                                                      * np.cos(3.0 * lev)) for iq in range(nq)], axis=-1)))
        fv.set_tracers(q0)
        del q0
    zero = np.zeros_like(st[0]["delp"])
    fv.dc.set_state([s_["u"] for s_ in st], [s_["v"] for s_ in st], [s_.get("w", zero) for s_ in st], [s_["delp"] for s_ in st],
                    [s_["pt"] for s_ in st], [s_.get("delz", bd.zeros("CC", npz)) for s_ in st], [s_["phis"] for s_ in st])
    del st
    areas = [np.asarray(g.m["area"])[c] for g in gs]

    def diag(day):
        d = fv.dc.d
        dp, u, v = d["delp"].download(), d["u"].download(), d["v"].download()
        ps = [ak[0] + np.sum(x[c], axis=2) for x in dp]
        mass = float(sum(np.sum(p_ * ar) for p_, ar in zip(ps, areas)))
        tr = {}
        if nq:
            q = d["q"].download()
            tm = [float(sum(np.sum(q_[c][..., iq] * x[c] * ar[:, :, None]) for q_, x, ar in zip(q, dp, areas))) for iq in range(nq)]
            tr = {"tracer_mass": tm, "q_min": float(min(q_[c].min() for q_ in q)), "q_max": float(max(q_[c].max() for q_ in q))}
        return {**tr, "day": day, "mass": mass, "ps_min_hPa": float(min(p_.min() for p_ in ps)) / 100.0,
                "ps_max_hPa": float(max(p_.max() for p_ in ps)) / 100.0,
                "u_max": float(max(np.abs(bd.view(x, "U", bd.is_, bd.ie, bd.js, bd.je + 1)).max() for x in u)),
                "v_max": float(max(np.abs(bd.view(x, "V", bd.is_, bd.ie + 1, bd.js, bd.je)).max() for x in v)),
                "finite": bool(all(np.isfinite(x[c]).all() for x in dp))}
    out = [diag(0.0)]
    fv.step(a.dt_atmos)
    torch.cuda.synchronize()
    nsteps = int(round(a.days * 86400.0 / a.dt_atmos))
    every = int(round(43200.0 / a.dt_atmos))
    t0 = time.perf_counter()
    for n in range(2, nsteps + 1):
        fv.step(a.dt_atmos)
        if n % every == 0:
            torch.cuda.synchronize()
            out.append(diag(n * a.dt_atmos / 86400.0))
            if not out[-1]["finite"]:
                break
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    m0 = out[0]["mass"]
    print(json.dumps({"flags": prod or "reference defaults",
                      "config": f"C{nx} L{npz} {'hydrostatic' if hyd else 'nonhydrostatic'} JW (test_case 13), dt_atmos {a.dt_atmos} s, k_split {a.k_split}, n_split {a.n_split}, "
                                f"whole sphere on one GPU, {str(nq) + ' tracers, ' if nq else ''}eager launches", "days": a.days, "steps": nsteps, "wall_s": wall,
                      "sypd": a.days / 365.0 / (wall / 86400.0), "mass_drift_rel": abs(out[-1]["mass"] - m0) / m0,
                      "nq": nq, "tracer_mass_drift_rel_max": (max(abs(b_ - a_) / abs(a_) for a_, b_ in zip(out[0]["tracer_mass"], out[-1]["tracer_mass"]))
                                                              if nq else None),
                      "ps_min_hPa_final": out[-1]["ps_min_hPa"], "build_id": L.build_id(), "series": out}))
    mctx.close()


if __name__ == "__main__":
    main()
