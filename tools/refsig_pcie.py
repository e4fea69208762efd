"""The PCIe-inclusive rate of the drop-in with the reference's argument list (fv3_dyn_core_mod: host arrays in, host arrays out on every
call): one 384 x 384 x 127 nonhydrostatic dyn_core call of n_split = 5 through fv3_solo_refsig, eager (every array copied in and out
every call) and with the lazy host-address registry (arrays nobody touched stay on the device), beside the resident Python host's time
for the same call.  Prints one JSON object.  It is never bench.py's `value` (that is the resident path)."""
import json
import os
import re
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import fortran_host as F
    from gfdl_atmos_cubed_sphere_amd import lib as L
    nx = int(sys.argv[1]) if len(sys.argv) > 1 else 384
    npz = int(sys.argv[2]) if len(sys.argv) > 2 else 127
    prod = L.load()
    out = {"shape": [nx, nx, npz], "n_split": 5, "build_id": L.build_id()}
    cells = nx * nx * npz
    for name, reg in (("eager", False), ("lazy_registry", True)):
        with tempfile.TemporaryDirectory() as d:
            t0 = time.perf_counter()
            log = F.check_fortran_refsig(prod, d, nx=nx, ny=nx, npz=npz, n_split=5, nsteps=4, bdt=5.0, registry=reg)
            m = re.search(r"seconds per dyn_core call \(host arrays in and out\) =\s*([0-9.Ee+-]+)", log)
            s = float(m.group(1))
            out[name] = {"s_per_dyn_core_call": s, "cell_updates_per_s": cells * 5 / s, "ms_per_substep": s / 5 * 1e3,
                         "check_wall_s": time.perf_counter() - t0}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
