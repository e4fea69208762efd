"""Print the observed error of d_sw against the oracle on the GPU for several schemes (expected: exactly 0.0): a check
that the shared-reciprocal divisions and the L2 atomics of the fused kernels leave the results bit-identical.
usage (GPU box): python tools/exact_check.py"""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import parity_common as P
from gfdl_atmos_cubed_sphere_amd import lib as L
prod = L.load()
for hyd in (False, True):
    for hord in (10, 8, 5, -5, 6):
        w = P.check_d_sw(prod, nx=130, ny=64, npz=3, perturb=False, hydrostatic=hyd,
                         par_over=dict(hord_dp=hord, hord_tm=hord, hord_vt=hord, hord_mt=max(hord, 5)))
        print("uniform hyd", hyd, "hord", hord, "worst", max(w.values()), {k: v for k, v in w.items() if v > 0})
w = P.check_d_sw(prod, nx=130, ny=100, npz=3)
print("general", max(w.values()), {k: v for k, v in w.items() if v > 0})
