// spmd.h -- the wavefront as the unit of work: one 64-lane wavefront owns a strip of 64 consecutive
// i-points and marches along j.  x-neighbours are reached with DPP wavefront shifts
// (v_mov_b32_dpp wave_shr:1 / wave_shl:1, no LDS, no barrier), y-neighbours live in registers.
//
// Kernels written against this header use the lane-value type `vd` (a double per lane) and the
// lane-predicate type `vb`.  In the product build (hipcc, gfx950) vd IS double and vb IS bool: the
// code is ordinary per-lane SIMT code.  Under -DFV3_HOST_EMU (tests/hostemu only) vd is a 64-element
// array with element-wise operators, so the very same kernel source runs wave-by-wave on the CPU as
// a logic-checking harness; it is never a product path.
//
// All control flow in such kernels must be wave-uniform (loop counters, level parameters); lane-
// dependent choices go through vsel().
#pragma once

#include "fv3_common.h"

namespace fv3 {

constexpr int kW = 64;  // lanes per wavefront (CDNA)

#ifdef FV3_HOST_EMU
// ------------------------------------------------------------------------------------ host emulation
struct vb {
  bool v[kW];
};
struct vd {
  double v[kW];
  vd() {}
  vd(double s) {
    for (int l = 0; l < kW; l++) v[l] = s;
  }
};
#define FV3_VOP2(op)                                                   \
  inline vd operator op(const vd &a, const vd &b) {                    \
    vd r;                                                              \
    for (int l = 0; l < kW; l++) r.v[l] = a.v[l] op b.v[l];            \
    return r;                                                          \
  }                                                                    \
  inline vd operator op(const vd &a, double b) {                       \
    vd r;                                                              \
    for (int l = 0; l < kW; l++) r.v[l] = a.v[l] op b;                 \
    return r;                                                          \
  }                                                                    \
  inline vd operator op(double a, const vd &b) {                       \
    vd r;                                                              \
    for (int l = 0; l < kW; l++) r.v[l] = a op b.v[l];                 \
    return r;                                                          \
  }
FV3_VOP2(+)
FV3_VOP2(-)
FV3_VOP2(*)
FV3_VOP2(/)
#undef FV3_VOP2
inline vd operator-(const vd &a) {
  vd r;
  for (int l = 0; l < kW; l++) r.v[l] = -a.v[l];
  return r;
}
#define FV3_VCMP(op)                                                   \
  inline vb operator op(const vd &a, const vd &b) {                    \
    vb r;                                                              \
    for (int l = 0; l < kW; l++) r.v[l] = a.v[l] op b.v[l];            \
    return r;                                                          \
  }                                                                    \
  inline vb operator op(const vd &a, double b) {                       \
    vb r;                                                              \
    for (int l = 0; l < kW; l++) r.v[l] = a.v[l] op b;                 \
    return r;                                                          \
  }
FV3_VCMP(<)
FV3_VCMP(>)
FV3_VCMP(<=)
FV3_VCMP(>=)
#undef FV3_VCMP
inline vb operator&&(const vb &a, const vb &b) {
  vb r;
  for (int l = 0; l < kW; l++) r.v[l] = a.v[l] && b.v[l];
  return r;
}
inline vb operator||(const vb &a, const vb &b) {
  vb r;
  for (int l = 0; l < kW; l++) r.v[l] = a.v[l] || b.v[l];
  return r;
}
inline vb operator!(const vb &a) {
  vb r;
  for (int l = 0; l < kW; l++) r.v[l] = !a.v[l];
  return r;
}
inline vd vsel(const vb &m, const vd &a, const vd &b) {
  vd r;
  for (int l = 0; l < kW; l++) r.v[l] = m.v[l] ? a.v[l] : b.v[l];
  return r;
}
inline vb vselb(const vb &m, const vb &a, const vb &b) {
  vb r;
  for (int l = 0; l < kW; l++) r.v[l] = m.v[l] ? a.v[l] : b.v[l];
  return r;
}
inline vd vmin(const vd &a, const vd &b) {
  vd r;
  for (int l = 0; l < kW; l++) r.v[l] = dmin(a.v[l], b.v[l]);
  return r;
}
inline vd vmax(const vd &a, const vd &b) {
  vd r;
  for (int l = 0; l < kW; l++) r.v[l] = dmax(a.v[l], b.v[l]);
  return r;
}
inline vd vabs(const vd &a) {
  vd r;
  for (int l = 0; l < kW; l++) r.v[l] = fabs(a.v[l]);
  return r;
}
inline vd vsign(const vd &a, const vd &b) {  // Fortran sign(a, b)
  vd r;
  for (int l = 0; l < kW; l++) r.v[l] = fsign(a.v[l], b.v[l]);
  return r;
}
inline vd vsqrt(const vd &a) {
  vd r;
  for (int l = 0; l < kW; l++) r.v[l] = sqrt(a.v[l]);
  return r;
}
// value of lane l-1 (lane 0 receives 0) / lane l+1 (lane 63 receives 0)
inline vd shr1(const vd &a) {
  vd r;
  r.v[0] = 0.;
  for (int l = 1; l < kW; l++) r.v[l] = a.v[l - 1];
  return r;
}
inline vd shl1(const vd &a) {
  vd r;
  r.v[kW - 1] = 0.;
  for (int l = 0; l < kW - 1; l++) r.v[l] = a.v[l + 1];
  return r;
}
// lane index clamped to [lmin, lmax]: loads through it are always in bounds, lanes outside the range
// receive a copy of the nearest valid element (never used for a value that is kept)
struct vl {
  int v[kW];
};
inline vl make_lanes(int lmin, int lmax) {
  vl r;
  for (int l = 0; l < kW; l++) r.v[l] = l < lmin ? lmin : (l > lmax ? lmax : l);
  return r;
}
// p[off + clamped lane]
inline vd vload(const double *p, long off, const vl &li) {
  vd r;
  for (int l = 0; l < kW; l++) r.v[l] = p[off + li.v[l]];
  return r;
}
inline void vstore(double *p, long off, const vd &x, int lmin, int lmax) {
  for (int l = 0; l < kW; l++)
    if (l >= lmin && l <= lmax) p[off + l] = x.v[l];
}
inline void vstore_nt(double *p, long off, const vd &x, int lmin, int lmax) { vstore(p, off, x, lmin, lmax); }
// correctly rounded a / b with a shared reciprocal (see the device version); the harness just divides
inline vd vrecip(const vd &b) {
  vd r;
  for (int l = 0; l < kW; l++) r.v[l] = 1. / b.v[l];
  return r;
}
inline vd vdiv_r(const vd &a, const vd &b, const vd &) { return a / b; }
inline void vaccum(double *p, long off, const vd &x, int lmin, int lmax) {
  for (int l = 0; l < kW; l++)
    if (l >= lmin && l <= lmax) p[off + l] = p[off + l] + x.v[l];
}
// predicate: lane in [l0, l1]
inline vb lane_mask(int l0, int l1) {
  vb r;
  for (int l = 0; l < kW; l++) r.v[l] = l >= l0 && l <= l1;
  return r;
}

#else
// ------------------------------------------------------------------------------------------ gfx950
using vd = double;
using vb = bool;

__device__ __forceinline__ vd vsel(vb m, vd a, vd b) { return m ? a : b; }
__device__ __forceinline__ vb vselb(vb m, vb a, vb b) { return m ? a : b; }
__device__ __forceinline__ vd vmin(vd a, vd b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ vd vmax(vd a, vd b) { return __builtin_fmax(a, b); }
__device__ __forceinline__ vd vabs(vd a) { return __builtin_fabs(a); }
__device__ __forceinline__ vd vsign(vd a, vd b) { return __builtin_copysign(a, b); }
__device__ __forceinline__ vd vsqrt(vd a) { return sqrt(a); }

// DPP wavefront shifts (GFX9 DPP_WF_SR1 = 0x138, DPP_WF_SL1 = 0x130): two v_mov_b32_dpp per double.
// bound_ctrl = true -> the lane without a source (0 resp. 63) receives 0.
__device__ __forceinline__ vd shr1(vd a) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(a), 0x138, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(a), 0x138, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ vd shl1(vd a) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(a), 0x130, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(a), 0x130, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
using vl = unsigned;  // clamped lane index as a byte offset
__device__ __forceinline__ vl make_lanes(int lmin, int lmax) {
  const int l = (int)(threadIdx.x & (kW - 1));
  return (unsigned)(l < lmin ? lmin : (l > lmax ? lmax : l)) * 8u;
}
// uniform base (SGPR pair) + per-lane unsigned 32-bit byte offset: global_load_dwordx2 v, v_off, s[base]
__device__ __forceinline__ vd vload(const double *p, long off, vl li) {
  return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(p + off) + li);
}
__device__ __forceinline__ void vstore(double *p, long off, vd x, int lmin, int lmax) {
  const int l = (int)(threadIdx.x & (kW - 1));
  if (l >= lmin && l <= lmax) (p + off)[l] = x;
}
// streaming store (global_store ... nt): for kernels that write many more rows than they re-read, so that the output
// does not push the input rows of the neighbouring wavefronts out of L2 (measured on the fused transport: -7 %)
// Division by a denominator that several numerators share (ra_x, ra_y, the new delp of a cell): y = RN(1/b) once
// (v_rcp_f64 + two Newton steps, the sequence the compiler's own fdiv expansion uses), then per numerator
// q0 = RN(a*y), r = a - b*q0 (exact in an fma), q = RN(q0 + r*y).  With y correctly rounded q is the correctly rounded
// quotient (Markstein 1990) -- the IEEE result of a / b for operands in the normal range (no v_div_scale / v_div_fixup
// rescue here: the operands are areas, pressure thicknesses and tracer masses) -- at 3 instructions per numerator instead
// of 11.
__device__ __forceinline__ vd vrecip(vd b) {
  double y = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-b, y, 1.0);
  return __builtin_fma(y, e, y);
}
__device__ __forceinline__ vd vdiv_r(vd a, vd b, vd y) {
  const double q0 = a * y;
  const double r = __builtin_fma(-b, q0, a);
  return __builtin_fma(r, y, q0);
}
// p[off + lane] += x as one read-modify-write in L2 (global_atomic_add_f64, no return value): the accumulators of the
// flux capacitors (cx, cy, mfx, mfy) need no registers for their old value and no load.  Every element is updated by
// exactly one lane of one wavefront, so the result is the plain IEEE sum.
__device__ __forceinline__ void vaccum(double *p, long off, vd x, int lmin, int lmax) {
  const int l = (int)(threadIdx.x & (kW - 1));
  if (l >= lmin && l <= lmax) unsafeAtomicAdd(p + off + l, x);
}
__device__ __forceinline__ void vstore_nt(double *p, long off, vd x, int lmin, int lmax) {
  const int l = (int)(threadIdx.x & (kW - 1));
  if (l >= lmin && l <= lmax) __builtin_nontemporal_store(x, p + off + l);
}
#endif

#ifndef FV3_HOST_EMU
__device__ __forceinline__ vb lane_mask(int l0, int l1) {
  const int l = (int)(threadIdx.x & (kW - 1));
  return l >= l0 && l <= l1;
}
#endif

FV3_HD vd vmin3(const vd &a, const vd &b, const vd &c) { return vmin(vmin(a, b), c); }
FV3_HD vd vmax3(const vd &a, const vd &b, const vd &c) { return vmax(vmax(a, b), c); }

}  // namespace fv3
