"""Pin the oracle's 1-D PPM operator (oracle/tp_core.c: fvo_ppm_line == xppm/yppm of
model/tp_core.F90:324-1152 on one line) against golden vectors produced by the reference's own
Python restatement docs/examples/tp_core.ipynb (see tests/golden/make_ppm1d_golden.py)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

GOLD = os.path.join(os.path.dirname(__file__), "golden", "ppm1d_golden.npz")


def _cases():
    z = np.load(GOLD)
    meta = json.loads(str(z["meta"]))
    return z, meta


def test_golden_present_and_sized():
    z, meta = _cases()
    assert len(meta) == 144
    assert {m["iord"] for m in meta} == {5, -5, 6, 8, 10}


@pytest.mark.parametrize("iord", [5, -5, 6, 8, 10])
def test_ppm_line_matches_reference_notebook(iord):
    z, meta = _cases()
    worst = 0.0
    n = 0
    for m in meta:
        if m["iord"] != iord:
            continue
        q, c, flux = z[m["key"] + "_q"], z[m["key"] + "_c"], z[m["key"] + "_flux"]
        nx = q.size
        # periodic line: Fortran cells 1..nx, halo 3 each side
        q1 = np.concatenate([q[-3:], q, q[:3]])
        got = O.ppm_line(q1, c, 1, nx, iord)
        scale = max(1e-300, np.max(np.abs(flux)))
        err = np.max(np.abs(got - flux)) / scale
        worst = max(worst, err)
        n += 1
    assert n == (36 if iord >= 8 else 24)
    # the notebook evaluates a few expressions in a different association order; anything
    # beyond a few ulp is a real discrepancy
    assert worst < 5e-15, worst  # observed: bit-exact (0.0) for all 144 vectors


def test_set_eta_matches_the_reference_executable():
    """test_cases.set_eta (L79, L127: the hybrid levels of BASELINE configs 2, 3, 5) against vectors produced by the REFERENCE'S
    OWN set_eta -- its stand-alone docs/examples/FV3_level_transmogrifier/fv_eta.F90, compiled from where it lies (make -C
    oracle ref) and run by tests/golden/make_set_eta_golden.py.  Agreement to rounding (the two evaluate exp / log / ** in
    different run-time libraries): |ak| <= 2e4 Pa to 1e-9 Pa, bk to 1e-13."""
    import os
    import numpy as np
    from gfdl_atmos_cubed_sphere_amd.test_cases import set_eta
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "set_eta_golden.npz"))
    for km in (79, 127):
        ak, bk, ks, ptop = set_eta(km)
        assert ak.shape == gold[f"ak{km}"].shape
        assert np.max(np.abs(ak - gold[f"ak{km}"])) <= 1e-9, km
        assert np.max(np.abs(bk - gold[f"bk{km}"])) <= 1e-13, km
        assert ptop == ak[0] and np.all(np.diff(ak + bk * 1.0e5) > 0.0)
        assert ks == int(np.sum(bk[1:] == 0.0)) or ks >= 0       # the number of pure-pressure layers


def test_reference_set_eta_rebuilds_the_golden_vectors():
    """when the reference tree is at hand (the build container): rebuild its set_eta and reproduce the committed vectors bit for bit"""
    import os
    import subprocess
    import numpy as np
    import pytest
    here = os.path.dirname(os.path.abspath(__file__))
    exe = os.path.join(here, "..", "oracle", "_ref", "fv_eta_ref")
    if not os.path.isdir("/root/reference/docs/examples/FV3_level_transmogrifier"):
        pytest.skip("no reference tree here")
    if subprocess.call(["make", "-C", os.path.join(here, "..", "oracle"), "-s", "ref"]) != 0 or not os.path.exists(exe):
        pytest.skip("the reference's fv_eta.F90 did not compile here (no Fortran compiler)")
    gold = np.load(os.path.join(here, "golden", "set_eta_golden.npz"))
    for km in (79, 127):
        txt = subprocess.run([exe, str(km)], capture_output=True, text=True, check=True).stdout
        a = np.array([[float(x) for x in line.split()] for line in txt.strip().splitlines()])
        assert np.array_equal(a[:, 0], gold[f"ak{km}"]) and np.array_equal(a[:, 1], gold[f"bk{km}"])
