"""The two lanes of the cubed-sphere c_sw / d_sw (side stream beside the marching kernels, fv3_api.hip dsw_cubed / csw_cubed) against
the one-lane order (FV3_MI355X_SIDE_STREAM=0): every output of REPS pairs on one gnomonic face must be equal bit for bit, and the
wall time of the pair is printed for both.  NPX, NPZ, REPS, NH, PROD from the environment."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


_STATE = {}


def run(side, npx, npz, nh, prod, reps, tile):
    from gfdl_atmos_cubed_sphere_amd.dyn_core import DynFlags, level_coefficients
    from gfdl_atmos_cubed_sphere_amd.lib import Context
    from gfdl_atmos_cubed_sphere_amd.synthetic import CSW_OUT, DSW_PAR
    import cubed_common as CC
    os.environ["FV3_MI355X_SIDE_STREAM"] = "2" if side else "0"   # 2: the two lanes whatever the size of the face
    key = (npx, npz, nh)
    if key not in _STATE:
        _STATE[key] = CC.global_state(npx, npz, hydrostatic=not nh)
    cs, gs, st = _STATE[key]
    g = gs[tile]
    ctx = Context(g, npz)
    fl = DynFlags(hydrostatic=not nh, **(dict(do_vort_damp=True, vtdm4=0.06, nord=3, d_con=1.0, dddmp=0.5) if prod else {}))
    ctx.dsw_levels(level_coefficients(npz, fl))
    d = {k: ctx.from_host(v) for k, v in st[tile].items()}
    for n, kind in CSW_OUT:
        d[n] = ctx.zeros(kind, npz)
    for n, kind in (("mfx", "FX"), ("mfy", "FY"), ("cx", "CX"), ("cy", "CY"), ("crx", "CX"), ("cry", "CY"), ("xfx", "CX"), ("yfx", "CY")):
        d[n] = ctx.zeros(kind, npz)
    out = {n: ctx.zeros(kind, npz) for n, kind in (("delp_out", "A"), ("pt_out", "A"), ("u_out", "U"), ("v_out", "V"), ("w_out", "A"),
                                                   ("heat_s", "CC"), ("diss_e", "CC"))}
    par = dict(DSW_PAR)
    par.update(dt=30.0, hydrostatic=int(not nh), use_cond=0, dddmp=fl.dddmp)
    hyd = not nh

    def pair():
        ctx.c_sw(d["delpc"], d["delp"], d["ptc"], d["pt"], d["u"], d["v"], d.get("w"), d["uc"], d["vc"], d["ua"], d["va"],
                 None if hyd else d["wc"], d["ut"], d["vt"], d["divg_d"], 1, 15.0, hyd)
        ctx.d_sw(par, d["vt"], d["delp"], d["pt"], d["u"], d["v"], d.get("w"), d["uc"], d["vc"], d["ua"], d["va"], d["divg_d"],
                 d["mfx"], d["mfy"], d["cx"], d["cy"], d["crx"], d["cry"], d["xfx"], d["yfx"], None, out["delp_out"], out["pt_out"],
                 out["u_out"], out["v_out"], None if hyd else out["w_out"], None, out["heat_s"], out["diss_e"])
    names = ["uc", "vc", "ua", "va", "ut", "vt", "divg_d", "delpc", "ptc", "mfx", "mfy", "cx", "cy", "crx", "cry", "xfx", "yfx"] + ([] if hyd else ["wc"])
    res = []
    for r in range(reps):
        for n in ("mfx", "mfy", "cx", "cy"):
            d[n].upload(np.zeros(d[n].shape))
        pair()
        ctx.sync()
        if r in (0, reps - 1):
            res.append({**{n: d[n].download() for n in names}, **{n: v.download() for n, v in out.items() if not (hyd and n == "w_out")}})
    t0 = time.perf_counter()
    for _ in range(10):
        pair()
    ctx.sync()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    ctx.close()
    return res, ms


def main():
    npx, npz = int(os.environ.get("NPX", 97)), int(os.environ.get("NPZ", 32))
    nh, prod, reps = os.environ.get("NH", "1") == "1", os.environ.get("PROD", "0") == "1", int(os.environ.get("REPS", 6))
    bad = 0
    if os.environ.get("FV3_LANES_ONLY_TWO") == "1":   # (tools/lanes_timeline.sh: only the two-lane run, for a kernel trace)
        _, ms_b = run(True, npx, npz, nh, prod, reps, 0)
        print(f"two lanes {ms_b:.3f} ms per pair")
        return
    for tile in (0, 3):
        a, ms_a = run(False, npx, npz, nh, prod, reps, tile)
        b, ms_b = run(True, npx, npz, nh, prod, reps, tile)
        for ra, rb in zip(a, b):
            for n in ra:
                if not np.array_equal(ra[n], rb[n]):
                    bad += 1
                    print("DIFFERENT", tile, n, float(np.max(np.abs(ra[n] - rb[n]))))
        print(f"tile {tile}: one lane {ms_a:.3f} ms, two lanes {ms_b:.3f} ms per pair (C{npx - 1} L{npz}, nh={nh}, prod={prod})")
    print("lanes_check", "FAILED" if bad else "ok: every output equal bit for bit")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
