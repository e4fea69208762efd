"""Golden vectors for set_eta from the REFERENCE ITSELF: oracle/_ref/fv_eta_ref is the reference's own stand-alone fv_eta
(/root/reference/docs/examples/FV3_level_transmogrifier/fv_eta.F90 + fv_eta.h, compiled where it lies by `make -C oracle ref`
with amdflang -fdefault-real-8) behind a ten-line driver.  Run in the build container (the GPU box has no /root/reference):

    make -C oracle ref && python tests/golden/make_set_eta_golden.py

-> tests/golden/set_eta_golden.npz: ak, bk of every level count that gfdl_atmos_cubed_sphere_amd/test_cases.py::set_eta
restates (79 and 127: BASELINE configs 2, 3, 5).  Data only."""
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(HERE, "..", "..", "oracle", "_ref", "fv_eta_ref")


def main():
    out = {}
    for km in (79, 127):
        txt = subprocess.run([EXE, str(km)], capture_output=True, text=True, check=True).stdout
        a = np.array([[float(x) for x in line.split()] for line in txt.strip().splitlines()])
        assert a.shape == (km + 1, 2)
        out[f"ak{km}"], out[f"bk{km}"] = a[:, 0], a[:, 1]
    np.savez(os.path.join(HERE, "set_eta_golden.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
