// fv3_common.h -- shared definitions of the HIP kernels (gfx950).
//
// The tile kernels are written as functors `void operator()(bx, by, bz, tid, lds)`.  In the
// product build (hipcc) they run under the generic __global__ launcher in fv3_launch.h with
// kNT threads per workgroup.  Every stage of a kernel is a data-parallel loop
// `for (idx = tid; idx < n; idx += kNT)` separated by FV3_SYNC() barriers, so the same source
// can also be compiled by g++ with -DFV3_HOST_EMU (kNT = 1, barrier = no-op) -- that build
// exists ONLY under tests/hostemu as a kernel-logic test harness for the CPU-only container;
// the shipped library is HIP-only and has no CPU path.
#pragma once

#include <cmath>
#include <cstddef>
#include <cstdint>

#ifdef FV3_HOST_EMU
#define FV3_HD inline
#define FV3_D inline
#define FV3_SYNC() ((void)0)
#define FV3_SYNC_LDS() ((void)0)
constexpr int kNT = 1;
#else
#include <hip/hip_runtime.h>
#define FV3_HD __host__ __device__ __forceinline__
#define FV3_D __device__ __forceinline__
#define FV3_SYNC() __syncthreads()
// A workgroup barrier that orders the LDS traffic only: __syncthreads() also waits for every global load and STORE of the wavefront
// (s_waitcnt vmcnt(0)), i.e. a kernel that stores a field and stages the next one pays the write latency and then the read latency.
// For phases that exchange data through LDS alone (any global location is written and read back by the SAME thread, which the
// hardware keeps in program order) the stores may drain and prefetched loads stay in flight across the barrier.
#define FV3_SYNC_LDS()                                               \
  do {                                                               \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");  \
    __builtin_amdgcn_s_barrier();                                    \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");  \
  } while (0)
constexpr int kNT = 256;
#endif

// the thread's index, opaque to the optimizer from here on: addresses formed from it are formed again instead of living in registers
// as common subexpressions of a whole kernel (remap_fast.h)
#ifdef FV3_HOST_EMU
inline int fresh_tid(int x) { return x; }
#else
__device__ __forceinline__ int fresh_tid(int x) { asm volatile("" : "+v"(x)); return x; }
#endif

// the one exp / log of the column kernels and of their checker (see the header for why)
#define FV3M_FN FV3_HD
#include "../../include/fv3_math.h"

namespace fv3 {

// wavefronts per SIMD the "_2w" launchers of fv3_launch.h budget the registers of a tile functor for (amdgpu_waves_per_eu): 2 unless
// the functor says otherwise (remap_fast.h: the 5-levels-per-lane kernels fit three workgroups per CU)
template <class F>
struct tile_waves { static constexpr int value = 2; };

// every pressure power / log-pressure of the path goes through these two
FV3_HD double dexp(double x) { return fv3_exp(x); }
FV3_HD double dlog(double x) { return fv3_log(x); }

constexpr int NG = 3;  // halo width (tools/fv_mp_mod.F90:61)

// Device-side view of the domain + gridstruct (all pointers are device pointers).
struct Grid {
  int is, ie, js, je, isd, ied, jsd, jed;
  int npx, npy, npz;
  int nid, njd, nx, ny;  // nid = ied-isd+1, nx = ie-is+1
  int grid_type, do_diss_est, prevent_diss_cooling, stretched_grid;
  double lim_fac, da_min, da_min_c;
  const double *area, *rarea, *dxa, *dya, *rdxa, *rdya, *cosa_s, *rsin2, *f0;       // A
  const double *dx, *rdx, *dyc, *rdyc, *cosa_v, *sina_v, *rsin_v, *divg_u, *del6_u; // U
  const double *dy, *rdy, *dxc, *rdxc, *cosa_u, *sina_u, *rsin_u, *divg_v, *del6_v; // V
  const double *rarea_c, *fC, *cosa, *sina;                                          // B
  const double *sin_sg, *cos_sg;                                                     // A x 9
  // geometry mode, found by fv3_grid_upload from the metric arrays themselves: 0 = general; 1 = orthogonal (cosa* = 0,
  // sin* = rsin* = 1 everywhere, as fv_grid_utils.F90:427 sets them for grid_type >= 3); 2 = orthogonal and every length /
  // area term spatially constant (Cartesian doubly periodic, fv_grid_tools.F90:1202-1221): the constants are below
  int geom;
  double c_area, c_rarea, c_dxa, c_dya, c_rdxa, c_rdya, c_dx, c_rdx, c_dyc, c_rdyc, c_dy, c_rdy, c_dxc, c_rdxc;
  double c_divg_u, c_divg_v, c_del6_u, c_del6_v, c_rarea_c;

  // flat index of (i,j) (Fortran indices) in one k-slab of each stagger kind
  FV3_HD int iA(int i, int j) const { return (j - jsd) * nid + (i - isd); }
  FV3_HD int iU(int i, int j) const { return (j - jsd) * nid + (i - isd); }
  FV3_HD int iV(int i, int j) const { return (j - jsd) * (nid + 1) + (i - isd); }
  FV3_HD int iB(int i, int j) const { return (j - jsd) * (nid + 1) + (i - isd); }
  FV3_HD int iCX(int i, int j) const { return (j - jsd) * (nx + 1) + (i - is); }
  FV3_HD int iCY(int i, int j) const { return (j - js) * nid + (i - isd); }
  FV3_HD int iFX(int i, int j) const { return (j - js) * (nx + 1) + (i - is); }
  FV3_HD int iFY(int i, int j) const { return (j - js) * nx + (i - is); }
  FV3_HD int iCC(int i, int j) const { return (j - js) * nx + (i - is); }
  FV3_HD int iRX(int i, int j) const { return (j - jsd) * nx + (i - is); }
  FV3_HD int iRY(int i, int j) const { return (j - js) * nid + (i - isd); }
  // slab sizes
  FV3_HD size_t nA() const { return (size_t)nid * njd; }
  FV3_HD size_t nU() const { return (size_t)nid * (njd + 1); }
  FV3_HD size_t nV() const { return (size_t)(nid + 1) * njd; }
  FV3_HD size_t nB() const { return (size_t)(nid + 1) * (njd + 1); }
  FV3_HD size_t nCX() const { return (size_t)(nx + 1) * njd; }
  FV3_HD size_t nCY() const { return (size_t)nid * (ny + 1); }
  FV3_HD size_t nFX() const { return (size_t)(nx + 1) * ny; }
  FV3_HD size_t nFY() const { return (size_t)nx * (ny + 1); }
  FV3_HD size_t nCC() const { return (size_t)nx * ny; }
  FV3_HD size_t nRX() const { return (size_t)nx * njd; }
  FV3_HD size_t nRY() const { return (size_t)nid * ny; }
  // sin_sg(i,j,n), n = 1..9 (model/fv_grid_utils.F90:91-97)
  FV3_HD double sinsg(int i, int j, int n) const { return sin_sg[(n - 1) * nid * njd + iA(i, j)]; }
  FV3_HD double cossg(int i, int j, int n) const { return cos_sg[(n - 1) * nid * njd + iA(i, j)]; }
};

// ---- scalar helpers with the reference's semantics ----------------------------------------
// v_min_f64 / v_max_f64: identical to the Fortran min/max intrinsics for non-NaN data
FV3_HD double dmin(double a, double b) { return __builtin_fmin(a, b); }
FV3_HD double dmax(double a, double b) { return __builtin_fmax(a, b); }
FV3_HD double dmin3(double a, double b, double c) { return dmin(dmin(a, b), c); }
FV3_HD double dmax3(double a, double b, double c) { return dmax(dmax(a, b), c); }
// Fortran sign(a, b)
FV3_HD double fsign(double a, double b) { return copysign(fabs(a), b); }
// Fortran x**n, integer n >= 1
FV3_HD double ipow(double x, int n) {
  double r = x;
  for (int k = 1; k < n; k++) r = r * x;
  return r;
}

// Data-parallel loop of a workgroup over a W x H index box: (li, lj) are advanced incrementally
// (no division per iteration).  W, H should be compile-time constants.
#define FV3_TILE_FOR(W, H, li, lj)                                                              \
  for (int idx_ = tid, li = tid % (W), lj = tid / (W); idx_ < (W) * (H);                        \
       idx_ += kNT, li += kNT % (W), lj += kNT / (W), lj += (li >= (W)) ? 1 : 0, li -= (li >= (W)) ? (W) : 0)

// A 2-D tile in LDS (or any memory) addressed with global Fortran indices.
struct Tile {
  double *p;
  int i0, j0, pitch;  // element (i,j) lives at p[(j-j0)*pitch + (i-i0)]
  FV3_HD double &operator()(int i, int j) const { return p[(j - j0) * pitch + (i - i0)]; }
};

}  // namespace fv3
