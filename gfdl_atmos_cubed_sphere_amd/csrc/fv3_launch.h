// fv3_launch.h -- runtime shim: the generic tile-kernel launcher and memory helpers.
// Product build: HIP (gfx950).  tests/hostemu build (-DFV3_HOST_EMU): plain C++ loops over the
// workgroup grid with one "thread" per group -- a logic-checking harness, not a product path.
#pragma once

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "fv3_common.h"

namespace fv3 {

#ifdef FV3_HOST_EMU

using stream_t = void *;
struct Dim3 {
  unsigned x, y, z;
};

template <class F>
inline int launch(Dim3 grid, size_t lds_doubles, stream_t, const F &f) {
  std::vector<double> lds(lds_doubles + 1, 0.);
  for (unsigned z = 0; z < grid.z; z++)
    for (unsigned y = 0; y < grid.y; y++)
      for (unsigned x = 0; x < grid.x; x++) f((int)x, (int)y, (int)z, 0, lds.data());  // order irrelevant here
  return 0;
}
template <class F>
inline int launch_2w(Dim3 grid, size_t lds_doubles, stream_t s, const F &f) { return launch(grid, lds_doubles, s, f); }
// column kernels (one thread per column)
template <int W = 0, class F>
inline int launch_cols(Dim3 grid, stream_t s, const F &f, int = 0) { return launch(grid, 0, s, f); }
// wave functors: one call per wavefront, the functor's vd values are 64-lane arrays (spmd.h)
template <class F>
inline int launch_waves(int nwaves, stream_t, const F &f) {
  for (int w = 0; w < nwaves; w++) f(w);
  return 0;
}
// a kernel of several faces in ONE launch (fv3_group, fv3_api.hip): here simply one face after the other
constexpr int kGrpMax = 6;
template <int KIND, class F>
inline int launch_group(Dim3 grid, size_t lds_doubles, int a, stream_t s, const F *const *fs, int n) {
  for (int m = 0; m < n; m++) {
    int rc = 0;
    if constexpr (KIND == 0) rc = launch(grid, lds_doubles, s, *fs[m]);
    else if constexpr (KIND == 1) rc = launch_2w(grid, lds_doubles, s, *fs[m]);
    else rc = launch_waves(a, s, *fs[m]);
    if (rc) return rc;
  }
  return 0;
}
template <int W, class F>
inline int launch_group_cols(Dim3 grid, int lanes, stream_t s, const F *const *fs, int n) {
  for (int m = 0; m < n; m++)
    if (int rc = launch_cols<W>(grid, s, *fs[m], lanes)) return rc;
  return 0;
}
inline int rt_malloc(void **p, size_t n) {
  *p = std::malloc(n);
  return *p ? 0 : 1;
}
inline int rt_free(void *p) {
  std::free(p);
  return 0;
}
inline int rt_h2d(void *d, const void *s, size_t n, stream_t) {
  std::memcpy(d, s, n);
  return 0;
}
inline int rt_d2h(void *d, const void *s, size_t n, stream_t) {
  std::memcpy(d, s, n);
  return 0;
}
inline int rt_memset(void *d, int v, size_t n, stream_t) {
  std::memset(d, v, n);
  return 0;
}
inline int rt_d2d(void *d, const void *s, size_t n, stream_t) {
  std::memcpy(d, s, n);
  return 0;
}
inline int rt_sync(stream_t) { return 0; }
inline const char *rt_errstr(int) { return "host-emu error"; }
inline int rt_stream_create(stream_t *s) { *s = nullptr; return 0; }
inline int rt_stream_create_low(stream_t *s) { *s = nullptr; return 0; }
inline void rt_stream_destroy(stream_t) {}
inline void rt_stream_wait_event(stream_t, void *) {}
inline int rt_event_create(void **e) { *e = nullptr; return 0; }
inline void rt_event_record(void *, stream_t) {}
inline double rt_event_elapsed_ms(void *, void *) { return 0.; }
inline void rt_event_destroy(void *) {}

#else  // ---------------------------------------------------------------------------- HIP

using stream_t = hipStream_t;
using Dim3 = dim3;

template <class F>
__global__ void __launch_bounds__(kNT) tile_kernel(const F f) {
  extern __shared__ double fv3_lds[];
  // The hardware grid is (nk, nbx, nby): the level index k varies fastest over consecutively
  // dispatched workgroups, which land round-robin on the 8 XCDs -- so each XCD's L2 sees the
  // same (i,j) tile for ~nk/8 levels in a row and the 2-D metric terms / tile halos are re-used
  // from L2 instead of being re-fetched per level.  Functors still see (bx, by, bz=k).
  f((int)blockIdx.y, (int)blockIdx.z, (int)blockIdx.x, (int)threadIdx.x, fv3_lds);
}

template <class F>
inline int launch(Dim3 grid, size_t lds_doubles, stream_t s, const F &f) {
  const size_t bytes = lds_doubles * sizeof(double);
  if (bytes > 64 * 1024) {
    static thread_local bool done = false;  // per functor type (template instantiation)
    if (!done) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&tile_kernel<F>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
      if (e != hipSuccess) return (int)e;
      done = true;
    }
  }
  hipLaunchKernelGGL(tile_kernel<F>, dim3(grid.z, grid.x, grid.y), dim3(kNT), bytes, s, f);
  return (int)hipGetLastError();
}
// the same under a register budget of two wavefronts per SIMD (<= 256 VGPRs + AGPRs): kernels whose latency hiding needs the second
// workgroup of a CU more than the last registers (nh_fast.h)
template <class F>
__global__ void __launch_bounds__(kNT) __attribute__((amdgpu_waves_per_eu(tile_waves<F>::value, tile_waves<F>::value))) tile_kernel_2w(const F f) {
  extern __shared__ double fv3_lds[];
  f((int)blockIdx.y, (int)blockIdx.z, (int)blockIdx.x, (int)threadIdx.x, fv3_lds);
}
template <class F>
inline int launch_2w(Dim3 grid, size_t lds_doubles, stream_t s, const F &f) {
  const size_t bytes = lds_doubles * sizeof(double);
  if (bytes > 64 * 1024) {
    static thread_local bool done = false;
    if (!done) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&tile_kernel_2w<F>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
      if (e != hipSuccess) return (int)e;
      done = true;
    }
  }
  hipLaunchKernelGGL(tile_kernel_2w<F>, dim3(grid.z, grid.x, grid.y), dim3(kNT), bytes, s, f);
  return (int)hipGetLastError();
}
// Column kernels (one thread per (i,j) column, long serial k loops, no LDS): launched as 64-thread workgroups so that
// the ~2300 wavefronts of a 384 x 384 tile spread evenly over the 1024 SIMDs (256-thread groups would give the 256
// CUs 2 or 3 groups each).  The functor still sees the (group of 256, thread) numbering of the tile launcher.
// Their k loops are chains of dependent, data-dependent loads, so what bounds them is the number of wavefronts a SIMD can
// switch between, not lanes: with `lanes` < 64 only the first `lanes` threads of each 64-thread group take a column, which
// multiplies the wavefronts in flight by 64 / lanes (FV3_MI355X_COL_LANES, measured best value is the default).
// Workgroups go to the 8 XCDs in turn (linear workgroup index mod 8).  A column kernel reads its neighbours' columns (rows j - 1, j + 1
// are six or seven 64-column workgroups away): with chunk > 0 (= workgroups per XCD) workgroup b of the first 8 chunk takes the logical
// index (XCD of b) * chunk + b / 8, so an XCD works through one contiguous range of columns and the neighbour rows meet in its own L2
// (FV3_MI355X_COL_XCD=0: the plain order).  `face`: the group kernels' blockIdx.y, whose workgroups continue the round-robin.
__device__ __forceinline__ int col_block(int chunk, int face = 0) {
  const int b = (int)blockIdx.x;
  if (b >= chunk * 8) return b;
  const int xcd = (b + face * (int)gridDim.x) & 7;
  return xcd * chunk + (b >> 3);
}
inline int col_xcd() {
  static const int v = [] {
    const char *e = std::getenv("FV3_MI355X_COL_XCD");
    return e ? std::atoi(e) : 1;
  }();
  return v;
}
template <class F>
__global__ void __launch_bounds__(64) col_kernel(const F f, int lanes, int chunk) {
  if ((int)threadIdx.x >= lanes) return;
  const int vt = col_block(chunk) * lanes + (int)threadIdx.x;  // virtual thread = column slot
  f(vt >> 8, 0, 0, vt & 255, nullptr);
}
inline int col_lanes() {
  static const int v = [] {
    const char *e = std::getenv("FV3_MI355X_COL_LANES");
    const int n = e ? std::atoi(e) : 64;
    return (n == 16 || n == 32) ? n : 64;
  }();
  return v;
}
// the same under a register budget of W wavefronts per SIMD (512 / W VGPRs): the column kernels wait on memory in a loop
// that is sequential in k, so for some of them more wavefronts in flight are worth a few spilled registers
template <class F, int W>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(W, W))) col_kernel_w(const F f, int lanes, int chunk) {
  if ((int)threadIdx.x >= lanes) return;
  const int vt = col_block(chunk) * lanes + (int)threadIdx.x;
  f(vt >> 8, 0, 0, vt & 255, nullptr);
}
// lanes_req > 0: columns per wavefront of this launch (any value <= 64; the default is col_lanes())
template <int W = 0, class F>
inline int launch_cols(Dim3 grid, stream_t s, const F &f, int lanes_req = 0) {
  const int lanes = (lanes_req > 0 && lanes_req <= 64) ? lanes_req : col_lanes();
  const unsigned nb = (unsigned)(((size_t)grid.x * 256 + lanes - 1) / lanes);
  const int chunk = (col_xcd() && nb >= 64) ? (int)(nb >> 3) : 0;
  if constexpr (W > 0)
    hipLaunchKernelGGL((col_kernel_w<F, W>), dim3(nb), dim3(64), 0, s, f, lanes, chunk);
  else
    hipLaunchKernelGGL(col_kernel<F>, dim3(nb), dim3(64), 0, s, f, lanes, chunk);
  return (int)hipGetLastError();
}
// wave functors (spmd.h): independent wavefronts, 4 per workgroup, no LDS, no barriers.
// Workgroups are handed to the 8 XCDs round-robin (workgroup b runs on XCD b % 8).  With chunk > 0 (= workgroups per
// XCD) workgroup b takes the logical index (b % 8) * chunk + b / 8, so each XCD works through one contiguous range of
// wavefront indices: the strips / segments that share halo rows and cache lines are neighbours in that order and meet
// in the same L2 instead of in eight different ones.
constexpr int kXcds = 8;
__device__ __forceinline__ int wave_index(int chunk) {
  const int b = (int)blockIdx.x;
  const int lb = chunk > 0 ? (b % kXcds) * chunk + b / kXcds : b;
  return lb * (kNT / 64) + (int)(threadIdx.x >> 6);
}
inline int xcd_remap() {
  static const int v = [] {
    const char *e = std::getenv("FV3_MI355X_XCD_REMAP");
    return e ? std::atoi(e) : 1;
  }();
  return v;
}
template <class F>
__global__ void __launch_bounds__(kNT) wave_kernel(const F f, int nwaves, int chunk) {
  // readfirstlane: tell the compiler the wave index is uniform, so everything derived from it
  // (level, strip, row offsets) lives in SGPRs and loads use the scalar-base addressing form
  const int gid = __builtin_amdgcn_readfirstlane(wave_index(chunk));
  if (gid < nwaves) f(gid);
}
// the same with a register budget of two wavefronts per SIMD (<= 256 VGPRs) for functors that ask for it with
// `static constexpr int kTwoWavesPerSimd = 1;` -- only worth it when the functor needs a few registers too many
template <class F>
__global__ void __launch_bounds__(kNT) __attribute__((amdgpu_waves_per_eu(2, 2))) wave_kernel_2w(const F f, int nwaves, int chunk) {
  const int gid = __builtin_amdgcn_readfirstlane(wave_index(chunk));
  if (gid < nwaves) f(gid);
}
// ... and of three (<= 168 VGPRs): `static constexpr int kThreeWavesPerSimd = 1;`
template <class F>
__global__ void __launch_bounds__(kNT) __attribute__((amdgpu_waves_per_eu(3, 3))) wave_kernel_3w(const F f, int nwaves, int chunk) {
  const int gid = __builtin_amdgcn_readfirstlane(wave_index(chunk));
  if (gid < nwaves) f(gid);
}
template <class F, class = void>
struct wants_three_waves : std::false_type {};
template <class F>
struct wants_three_waves<F, std::enable_if_t<(F::kThreeWavesPerSimd > 0)>> : std::true_type {};
template <class F, class = void>
struct wants_two_waves : std::false_type {};
template <class F>
struct wants_two_waves<F, std::enable_if_t<(F::kTwoWavesPerSimd > 0)>> : std::true_type {};

template <class F>
inline int launch_waves(int nwaves, stream_t s, const F &f) {
  const int wpb = kNT / 64;
  int nblocks = (nwaves + wpb - 1) / wpb, chunk = 0;
  if (xcd_remap() && nblocks >= 4 * kXcds) {
    chunk = (nblocks + kXcds - 1) / kXcds;
    nblocks = chunk * kXcds;
  }
  const dim3 grid((unsigned)nblocks);
  if constexpr (wants_three_waves<F>::value)
    hipLaunchKernelGGL(wave_kernel_3w<F>, grid, dim3(kNT), 0, s, f, nwaves, chunk);
  else if constexpr (wants_two_waves<F>::value)
    hipLaunchKernelGGL(wave_kernel_2w<F>, grid, dim3(kNT), 0, s, f, nwaves, chunk);
  else
    hipLaunchKernelGGL(wave_kernel<F>, grid, dim3(kNT), 0, s, f, nwaves, chunk);
  return (int)hipGetLastError();
}
// ---- a kernel of several faces in ONE launch (fv3_group, fv3_api.hip) -------------------------------------------------------------
// When one GPU holds several faces of the cube (six contexts), every kernel of the step is issued once per face: at C96 a face's pass /
// frame kernels are a few dozen workgroups for 256 CUs, and six of them one after the other cost six launch latencies for work that
// fills a sixth of the chip.  The group launchers take the functors of all faces BY VALUE in one kernel-argument block (six functors
// are 4-6 KB; the kernarg segment takes 60 KB, tools/probe/kernarg_probe.hip) and add the face as the slowest grid index.  The face is
// uniform per workgroup, so the functor's members are scalar loads from the kernarg segment at a dynamic offset -- no copy, no scratch.
constexpr int kGrpMax = 6;
template <class F>
struct FGroup {   // raw storage: the functors need no default constructor
  alignas(alignof(F)) unsigned char raw[kGrpMax * sizeof(F)];
  __device__ __forceinline__ const F &at(int m) const { return reinterpret_cast<const F *>(raw)[m]; }
};
template <class F>
__global__ void __launch_bounds__(kNT) tile_kernel_g(const FGroup<F> fg, int gy) {
  extern __shared__ double fv3_lds[];
  const int face = (int)blockIdx.z / gy, by = (int)blockIdx.z - face * gy;
  fg.at(face)((int)blockIdx.y, by, (int)blockIdx.x, (int)threadIdx.x, fv3_lds);
}
template <class F>
__global__ void __launch_bounds__(kNT) __attribute__((amdgpu_waves_per_eu(tile_waves<F>::value, tile_waves<F>::value))) tile_kernel_2w_g(const FGroup<F> fg, int gy) {
  extern __shared__ double fv3_lds[];
  const int face = (int)blockIdx.z / gy, by = (int)blockIdx.z - face * gy;
  fg.at(face)((int)blockIdx.y, by, (int)blockIdx.x, (int)threadIdx.x, fv3_lds);
}
template <class F>
__global__ void __launch_bounds__(64) col_kernel_g(const FGroup<F> fg, int lanes, int chunk) {
  if ((int)threadIdx.x >= lanes) return;
  const int vt = col_block(chunk, (int)blockIdx.y) * lanes + (int)threadIdx.x;
  fg.at((int)blockIdx.y)(vt >> 8, 0, 0, vt & 255, nullptr);
}
template <class F, int W>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(W, W))) col_kernel_w_g(const FGroup<F> fg, int lanes, int chunk) {
  if ((int)threadIdx.x >= lanes) return;
  const int vt = col_block(chunk, (int)blockIdx.y) * lanes + (int)threadIdx.x;
  fg.at((int)blockIdx.y)(vt >> 8, 0, 0, vt & 255, nullptr);
}
template <class F>
__global__ void __launch_bounds__(kNT) wave_kernel_g(const FGroup<F> fg, int nwaves, int chunk) {
  const int gid = __builtin_amdgcn_readfirstlane(wave_index(chunk));
  if (gid < nwaves) fg.at((int)blockIdx.y)(gid);
}
template <class F>
__global__ void __launch_bounds__(kNT) __attribute__((amdgpu_waves_per_eu(2, 2))) wave_kernel_2w_g(const FGroup<F> fg, int nwaves, int chunk) {
  const int gid = __builtin_amdgcn_readfirstlane(wave_index(chunk));
  if (gid < nwaves) fg.at((int)blockIdx.y)(gid);
}
template <class F>
__global__ void __launch_bounds__(kNT) __attribute__((amdgpu_waves_per_eu(3, 3))) wave_kernel_3w_g(const FGroup<F> fg, int nwaves, int chunk) {
  const int gid = __builtin_amdgcn_readfirstlane(wave_index(chunk));
  if (gid < nwaves) fg.at((int)blockIdx.y)(gid);
}
template <class F>
inline void fgroup_fill(FGroup<F> &fg, const F *const *fs, int n) {
  static_assert(std::is_trivially_copyable<F>::value, "kernel functors are plain data");
  for (int m = 0; m < kGrpMax; m++) std::memcpy((void *)(fg.raw + (size_t)m * sizeof(F)), (const void *)fs[m < n ? m : 0], sizeof(F));
}
// KIND: 0 = launch, 1 = launch_2w, 3 = launch_waves (a = nwaves); column kernels: launch_group_cols
template <int KIND, class F>
inline int launch_group(Dim3 grid, size_t lds_doubles, int a, stream_t s, const F *const *fs, int n) {
  FGroup<F> fg;
  fgroup_fill(fg, fs, n);
  if constexpr (KIND == 0 || KIND == 1) {
    const size_t bytes = lds_doubles * sizeof(double);
    if (bytes > 64 * 1024) {
      static thread_local bool done = false;
      if (!done) {
        hipError_t e = hipFuncSetAttribute(KIND ? reinterpret_cast<const void *>(&tile_kernel_2w_g<F>)
                                                : reinterpret_cast<const void *>(&tile_kernel_g<F>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return (int)e;
        done = true;
      }
    }
    const dim3 hw(grid.z, grid.x, grid.y * (unsigned)n);
    if constexpr (KIND == 1)
      hipLaunchKernelGGL(tile_kernel_2w_g<F>, hw, dim3(kNT), bytes, s, fg, (int)grid.y);
    else
      hipLaunchKernelGGL(tile_kernel_g<F>, hw, dim3(kNT), bytes, s, fg, (int)grid.y);
    return (int)hipGetLastError();
  } else {
    const int wpb = kNT / 64;
    int nblocks = (a + wpb - 1) / wpb, chunk = 0;
    if (xcd_remap() && nblocks >= 4 * kXcds) {
      chunk = (nblocks + kXcds - 1) / kXcds;
      nblocks = chunk * kXcds;
    }
    const dim3 hw((unsigned)nblocks, (unsigned)n);
    if constexpr (wants_three_waves<F>::value)
      hipLaunchKernelGGL(wave_kernel_3w_g<F>, hw, dim3(kNT), 0, s, fg, a, chunk);
    else if constexpr (wants_two_waves<F>::value)
      hipLaunchKernelGGL(wave_kernel_2w_g<F>, hw, dim3(kNT), 0, s, fg, a, chunk);
    else
      hipLaunchKernelGGL(wave_kernel_g<F>, hw, dim3(kNT), 0, s, fg, a, chunk);
    return (int)hipGetLastError();
  }
}
template <int W, class F>
inline int launch_group_cols(Dim3 grid, int lanes_req, stream_t s, const F *const *fs, int n) {
  FGroup<F> fg;
  fgroup_fill(fg, fs, n);
  const int lanes = (lanes_req > 0 && lanes_req <= 64) ? lanes_req : col_lanes();
  const unsigned nb = (unsigned)(((size_t)grid.x * 256 + lanes - 1) / lanes);
  const int chunk = (col_xcd() && nb >= 64) ? (int)(nb >> 3) : 0;
  if constexpr (W > 0)
    hipLaunchKernelGGL((col_kernel_w_g<F, W>), dim3(nb, (unsigned)n), dim3(64), 0, s, fg, lanes, chunk);
  else
    hipLaunchKernelGGL(col_kernel_g<F>, dim3(nb, (unsigned)n), dim3(64), 0, s, fg, lanes, chunk);
  return (int)hipGetLastError();
}
inline int rt_malloc(void **p, size_t n) { return (int)hipMalloc(p, n); }
inline int rt_free(void *p) { return (int)hipFree(p); }
inline int rt_h2d(void *d, const void *s, size_t n, stream_t st) {
  return (int)hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, st);
}
inline int rt_d2h(void *d, const void *s, size_t n, stream_t st) {
  return (int)hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, st);
}
inline int rt_memset(void *d, int v, size_t n, stream_t st) { return (int)hipMemsetAsync(d, v, n, st); }
inline int rt_d2d(void *d, const void *s, size_t n, stream_t st) {
  return (int)hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, st);
}
inline int rt_sync(stream_t st) { return (int)hipStreamSynchronize(st); }
inline const char *rt_errstr(int e) { return hipGetErrorString((hipError_t)e); }
inline int rt_stream_create(stream_t *s) { return (int)hipStreamCreateWithFlags(s, hipStreamNonBlocking); }
// a stream of the lowest priority: its workgroups are dispatched where the streams of normal priority leave room
inline int rt_stream_create_low(stream_t *s) {
  int least = 0, greatest = 0;
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return (int)hipStreamCreateWithFlags(s, hipStreamNonBlocking);
  return (int)hipStreamCreateWithPriority(s, hipStreamNonBlocking, least);
}
inline void rt_stream_destroy(stream_t s) { (void)hipStreamDestroy(s); }
inline void rt_stream_wait_event(stream_t s, void *e) { (void)hipStreamWaitEvent(s, (hipEvent_t)e, 0); }
inline int rt_event_create(void **e) { return (int)hipEventCreate((hipEvent_t *)e); }
inline void rt_event_record(void *e, stream_t s) { (void)hipEventRecord((hipEvent_t)e, s); }
inline double rt_event_elapsed_ms(void *a, void *b) {
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, (hipEvent_t)a, (hipEvent_t)b);
  return (double)ms;
}
inline void rt_event_destroy(void *e) { (void)hipEventDestroy((hipEvent_t)e); }

#endif

}  // namespace fv3
