"""Timing of the cubed-sphere (grid_type 0) kernels on one C384 L127 face: c_sw + d_sw (the pair), fv_tp_2d, and a whole
hydrostatic / nonhydrostatic substep through DynCore with six faces when --sphere is given.  Prints per-label launch
counts and times from the context's event profile, and the pair's cell-updates/s."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--npx", type=int, default=385)
    ap.add_argument("--npz", type=int, default=127)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--nh", action="store_true")
    ap.add_argument("--prod", action="store_true", help="nord = 3, do_vort_damp, vtdm4 = 0.06, d_con = 1, dddmp = 0.5")
    a = ap.parse_args()
    import torch
    from gfdl_atmos_cubed_sphere_amd.cubed_sphere import CubedSphere
    from gfdl_atmos_cubed_sphere_amd.dyn_core import DynFlags, level_coefficients
    from gfdl_atmos_cubed_sphere_amd.lib import Context
    from gfdl_atmos_cubed_sphere_amd.synthetic import CSW_OUT, DSW_PAR
    import cubed_common as CC
    npx, npz = a.npx, a.npz
    cs, gs, st = CC.global_state(npx, npz, hydrostatic=not a.nh)
    t = 0
    g, bd = gs[t], gs[t].bd
    ctx = Context(g, npz)
    fl = DynFlags(hydrostatic=not a.nh, **(dict(do_vort_damp=True, vtdm4=0.06, nord=3, d_con=1.0, dddmp=0.5) if a.prod else {}))
    ctx.dsw_levels(level_coefficients(npz, fl))
    d = {k: ctx.from_host(v) for k, v in st[t].items()}
    for n, kind in CSW_OUT:
        d[n] = ctx.zeros(kind, npz)
    for n, kind in (("mfx", "FX"), ("mfy", "FY"), ("cx", "CX"), ("cy", "CY"), ("crx", "CX"), ("cry", "CY"), ("xfx", "CX"), ("yfx", "CY")):
        d[n] = ctx.zeros(kind, npz)
    out = {n: ctx.zeros(kind, npz) for n, kind in (("delp_out", "A"), ("pt_out", "A"), ("u_out", "U"), ("v_out", "V"), ("w_out", "A"),
                                                   ("heat_s", "CC"), ("diss_e", "CC"))}
    par = dict(DSW_PAR)
    par.update(dt=30.0, hydrostatic=int(not a.nh), use_cond=0, dddmp=fl.dddmp)
    hyd = not a.nh

    def pair():
        ctx.c_sw(d["delpc"], d["delp"], d["ptc"], d["pt"], d["u"], d["v"], d.get("w"), d["uc"], d["vc"], d["ua"], d["va"],
                 None if hyd else d["wc"], d["ut"], d["vt"], d["divg_d"], 1, 15.0, hyd)
        ctx.d_sw(par, d["vt"], d["delp"], d["pt"], d["u"], d["v"], d.get("w"), d["uc"], d["vc"], d["ua"], d["va"], d["divg_d"],
                 d["mfx"], d["mfy"], d["cx"], d["cy"], d["crx"], d["cry"], d["xfx"], d["yfx"], None, out["delp_out"], out["pt_out"],
                 out["u_out"], out["v_out"], None if hyd else out["w_out"], None, out["heat_s"], out["diss_e"])
    for _ in range(3):
        pair()
    ctx.sync()
    ctx.profile(True)
    pair()
    ctx.sync()
    rep = ctx.profile_report()
    ctx.profile(False)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        pair()
    ctx.sync()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    cells = (npx - 1) ** 2 * npz
    print(json.dumps({"face": f"C{npx - 1}L{npz}", "hydrostatic": hyd, "pair_ms": ms, "cell_updates_per_s": cells / ms * 1e3,
                      "per_label": {k: [v[0], round(v[1], 4)] for k, v in rep.items()}}))


if __name__ == "__main__":
    main()
