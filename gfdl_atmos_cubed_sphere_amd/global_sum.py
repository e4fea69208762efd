"""g_sum(domain, p, ..., area, mode, reproduce=.true.) of the reference (model/fv_grid_utils.F90:2879-2925): the area-weighted
global sum behind the energy fixer (fv_mapz.F90:736-742), which the reference takes as
``mpp_global_sum(domain, p*area, flags=BITWISE_EFP_SUM)``.

That routine lives in FMS (github.com/NOAA-GFDL/FMS, mpp/mpp_efp.F90: mpp_reproducing_sum), a dependency that is not part of
the reference tree.  Its published algorithm -- the "extended fixed point" sum of Hallberg & Adcroft (2014, Parallel Computing
40, 140-143) -- is restated here: every addend is split into NUMINT = 6 integer digits of NUMBIT = 46 bits (radix 2**46, from
2**92 down to 2**-138), the digits are summed as integers (exact, so the result does not depend on the order of the addends nor
on how they are spread over ranks), carries are propagated, and the digits are turned back into a float from the most significant
one down.  Host code: the addends are the nx*ny column values of a face / block.

Without the FMS source at hand the last step (the order of the final floating-point additions) is the one thing that cannot be
checked against it; the digits themselves are exact, so any difference is at most one rounding of the global sum, i.e. ~1e-16
relative in dtmp, the single scalar that enters the state."""
from __future__ import annotations

import numpy as np

NUMBIT, NUMINT = 46, 6
_PREC = 1 << NUMBIT
_R = float(_PREC)
_PR = [_R * _R, _R, 1.0, 1.0 / _R, 1.0 / (_R * _R), 1.0 / (_R * _R * _R)]
_IPR = [1.0 / x for x in _PR]


def efp_digits(values) -> list[int]:
    """the NUMINT integer digits of sum(values), carries propagated so that every digit but the first is in [0, 2**46) (Python
    ints: exact)"""
    a = np.ascontiguousarray(values, dtype=np.float64).ravel()
    if not np.all(np.isfinite(a)):
        raise FloatingPointError("reproducing sum of a non-finite field")
    sgn = np.where(a < 0.0, -1, 1).astype(np.int64)
    rs = np.abs(a)
    if rs.size and float(rs.max()) >= _PR[0] * _R:      # the range of the format (FMS mpp_efp.F90 aborts with an overflow)
        raise OverflowError("reproducing sum: addend out of the range of the extended-fixed-point sum (|a| >= 2**138)")
    tot = []
    for i in range(NUMINT):
        iv = np.floor(rs * _IPR[i])
        rs = rs - iv * _PR[i]          # exact: iv * pr has at most 46 significant bits above pr
        d = iv.astype(np.int64) * sgn
        # |d| < 2**46 (i > 0): chunks of 2**16 addends stay inside int64, the chunk sums are added as Python ints
        tot.append(sum(int(x) for x in np.add.reduceat(d, np.arange(0, d.size, 1 << 16))) if d.size else 0)
    return _carry(tot)


def _carry(d: list[int]) -> list[int]:
    d = list(d)
    for i in range(NUMINT - 1, 0, -1):
        c = d[i] >> NUMBIT            # floor division: the remainder is in [0, 2**46)
        d[i] -= c << NUMBIT
        d[i - 1] += c
    return d


def digits_to_real(d: list[int]) -> float:
    r = 0.0
    for i in range(NUMINT):
        r = r + _PR[i] * float(d[i])
    return r


def reproducing_sum(parts, dist=None) -> float:
    """sum over every array in `parts` (one per face held by this process) and, with `dist`, over the ranks (the digits travel as
    int64 through all_reduce(SUM): exact)"""
    d = [0] * NUMINT
    for a in parts:
        d = [x + y for x, y in zip(d, efp_digits(a))]
    d = _carry(d)
    if abs(d[0]) >= (1 << 40):       # the same guard on one rank and on several, and in fv3_ordered_sum
        raise OverflowError("reproducing sum: leading digit too large (|sum| >= 2**132)")
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        import torch
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor(d, dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        d = _carry([int(x) for x in t.tolist()])
    return digits_to_real(d)


def g_sum(fields, areas, dist=None) -> float:
    """g_sum(..., mode = 0, reproduce = .true.): sum(p * area) over the compute domains; fields / areas: matching lists of (nx, ny)"""
    return reproducing_sum([np.asarray(p) * np.asarray(a) for p, a in zip(fields, areas)], dist)
