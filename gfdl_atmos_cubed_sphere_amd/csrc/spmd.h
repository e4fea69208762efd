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
// ---- branch-free rows (see the device section): a uniform row pointer, stores with a lane mask in the offset,
// an accumulation that adds zero on the lanes outside the mask
inline const double *urow(const double *p) { return p; }
inline double *urow(double *p) { return p; }
struct vm {
  int lmin, lmax;
};
inline vm make_mask(int lmin, int lmax) { return vm{lmin, lmax}; }
inline void vdrain_loads() {}
inline void vstore_b(double *p, long off, const vd &x, const vm &m, bool on = true) {
  if (on) vstore(p, off, x, m.lmin, m.lmax);
}
inline void vstore_b_nt(double *p, long off, const vd &x, const vm &m, bool on = true) {
  if (on) vstore(p, off, x, m.lmin, m.lmax);
}
inline void vaccum_z(double *p, long off, const vd &x, const vl &, const vm &m, bool on = true) {
  if (on) vaccum(p, off, x, m.lmin, m.lmax);
}
// m ? (value of lane l-1 of a) : b  -- on the device one v_cndmask_b32_dpp per half instead of a shift and a select
inline vd vsel_shr(const vb &m, const vd &a, const vd &b) { return vsel(m, shr1(a), b); }
inline vb vball(bool b) {   // a wave-uniform condition as a lane predicate
  vb r;
  for (int l = 0; l < kW; l++) r.v[l] = b;
  return r;
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
// clamped lane index as a byte offset.  `mutable`: vload passes the offset through an empty asm statement in place (see urow) -- the
// value never changes, but the optimizer may not hoist its 64-bit zero-extension out of the row loop
struct vl {
  mutable unsigned v;
};
__device__ __forceinline__ vl make_lanes(int lmin, int lmax) {
  const int l = (int)(threadIdx.x & (kW - 1));
  return vl{(unsigned)(l < lmin ? lmin : (l > lmax ? lmax : l)) * 8u};
}
// urow: the uniform row pointer made opaque to the optimizer (see "branch-free rows" below)
typedef const double __attribute__((address_space(1))) *fv3_gcptr;   // global address space: the laundered pointer must not decay to a flat one
typedef double __attribute__((address_space(1))) *fv3_gptr;
typedef const char __attribute__((address_space(1))) *fv3_gcbytes;
typedef char __attribute__((address_space(1))) *fv3_gbytes;
__device__ __forceinline__ fv3_gcptr urow(const double *p) {
  fv3_gcptr q = (fv3_gcptr)p;
  asm("" : "+s"(q));
  return q;
}
__device__ __forceinline__ fv3_gptr urow(double *p) {
  fv3_gptr q = (fv3_gptr)p;
  asm("" : "+s"(q));
  return q;
}
// uniform base (SGPR pair) + per-lane unsigned 32-bit byte offset: global_load_dwordx2 v, v_off, s[base]
__device__ __forceinline__ vd vload(const double *p, long off, const vl &li) {
  asm volatile("" : "+v"(li.v));  // keeps the zero-extension of the lane offset next to its use (a hoisted 64-bit copy defeats the saddr form)
  return *(fv3_gcptr)((fv3_gcbytes)urow(p + off) + li.v);
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
// ---- branch-free rows ------------------------------------------------------------------------------------------------------------
// A marching kernel whose row step contains no control flow lets the compiler count the memory operations in flight: the wait for the
// prefetched rows of the next step becomes s_waitcnt vmcnt(<stores of this step>) instead of vmcnt(0), i.e. the wavefront no longer waits
// for its own stores / atomics to be acknowledged before it may compute (measured: the fused transport kernel without its stores runs
// 28 % faster than with them, at 0.98x algorithmic traffic).  `if (lane in range) store` always keeps a branch (the compiler skips
// memory instructions under an empty EXEC), so the lane masks go where the hardware takes them without EXEC:
//  * urow(p): the uniform row pointer, opaque to the optimizer, so that loads / atomics keep the form
//    `global_load_dwordx2 v, v_lane_off, s[row:row+1]` (otherwise the loop-invariant part base + lane is hoisted into a VGPR pair per array
//    and every access pays a 64-bit VALU add);
//  * vm / vstore_b: stores through a buffer resource over the row; a lane outside the mask carries the byte offset 0x80000000, beyond
//    num_records, and the hardware drops its store (tools/probe/buf_oob.hip: the range check takes voffset + soffset against num_records);
//  * vaccum_z (an A / B variant, not used by the library's kernels: slower than the wave-uniform branch around vaccum, dsw_fused.h run_bf):
//    the L2 accumulation with the address clamped to valid elements and +0.0 on the lanes outside the mask (x + 0.0 == x for every x
//    that is not -0.0, and an accumulator that starts at +0.0 never becomes -0.0).
using vm = unsigned;  // byte offset of the lane, or 0x80000000 = masked
__device__ __forceinline__ vm make_mask(int lmin, int lmax) {
  const int l = (int)(threadIdx.x & (kW - 1));
  return (l >= lmin && l <= lmax) ? (unsigned)l * 8u : 0x80000000u;
}
typedef unsigned fv3_u2 __attribute__((ext_vector_type(2)));
// the resource covers the 64 lanes of ONE row (num_records = 512 bytes from the row pointer): it is rebuilt from the uniform row pointer
// at every store -- two scalar moves beside the pointer arithmetic a global store needs as well -- instead of holding four SGPRs per
// array for the whole loop (nine output arrays would not fit and spill into VGPR lanes)
// Before a branch-free row loop: every load issued so far has returned.  The compiler merges the memory operations pending on the two
// ways into the loop header; with the first row's loads still in flight on the way in, the header's wait would be the one of that
// way -- vmcnt(<loads>), which on the back edge means "all but the youngest stores done" -- whatever the back edge allows.
// (the two empty statements keep the loads above the wait and the loop's below it: the wait alone does not order them)
__device__ __forceinline__ void vdrain_loads() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) expcnt(7) lgkmcnt(15)
  asm volatile("" ::: "memory");
}
// `on` (wave-uniform): a row condition; off = num_records 0, every lane out of range -- a scalar select, no branch
// (readfirstlane: the record count must be a scalar whatever the optimizer made of the condition -- a resource with a VGPR word costs a
// waterfall loop per store)
__device__ __forceinline__ void vstore_b(double *p, long off, vd x, vm m, bool on = true) {
  const int nrec = __builtin_amdgcn_readfirstlane(on ? kW * 8 : 0);
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(fv3_u2, x), __builtin_amdgcn_make_buffer_rsrc(p + off, 0, nrec, 0x00020000), (int)m, 0, 0);
}
__device__ __forceinline__ void vstore_b_nt(double *p, long off, vd x, vm m, bool on = true) {
  const int nrec = __builtin_amdgcn_readfirstlane(on ? kW * 8 : 0);
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(fv3_u2, x), __builtin_amdgcn_make_buffer_rsrc(p + off, 0, nrec, 0x00020000), (int)m, 0, 2);  // aux bit 1 = nt
}
// p + off must be a valid row whatever `on` says (the lanes add +0.0 then)
__device__ __forceinline__ void vaccum_z(double *p, long off, vd x, const vl &clamp, vm m, bool on = true) {
  asm volatile("" : "+v"(clamp.v));
  const bool off_lane = (m & 0x80000000u) || !on;
  unsafeAtomicAdd((double *)(fv3_gptr)((fv3_gbytes)urow(p + off) + clamp.v), off_lane ? 0. : x);
}
#endif

#ifndef FV3_HOST_EMU
// m ? (value of lane l-1 of a, 0 in lane 0) : b.  The upwind selects of the x faces take the left cell's values through a wavefront
// shift; v_cndmask_b32 has a DPP form (VOP2: D = VCC ? src1 : dpp(src0)), which the compiler does not form by itself (its selects carry
// the mask in an SGPR pair, the VOP3 encoding, which has no DPP on gfx9): 2 instructions per double instead of 4.
// s_mov + s_nop: the two wait states a DPP read needs after the VALU write of its source (the hazard recognizer does not look into asm).
__device__ __forceinline__ vd vsel_shr(vb m, vd a, vd b) {
  const unsigned long long nm = __builtin_amdgcn_ballot_w64(!m);
  const int alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
  int rlo, rhi;
  asm("s_mov_b64 vcc, %4\n\ts_nop 0\n\t"
      "v_cndmask_b32_dpp %0, %2, %5, vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_cndmask_b32_dpp %1, %3, %6, vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
      : "=&v"(rlo), "=&v"(rhi)
      : "v"(alo), "v"(ahi), "s"(nm), "v"(blo), "v"(bhi)
      : "vcc");
  return __hiloint2double(rhi, rlo);
}
__device__ __forceinline__ vb vball(bool b) { return b; }
__device__ __forceinline__ vb lane_mask(int l0, int l1) {
  const int l = (int)(threadIdx.x & (kW - 1));
  return l >= l0 && l <= l1;
}
#endif

FV3_HD vd vmin3(const vd &a, const vd &b, const vd &c) { return vmin(vmin(a, b), c); }
FV3_HD vd vmax3(const vd &a, const vd &b, const vd &c) { return vmax(vmax(a, b), c); }

}  // namespace fv3
