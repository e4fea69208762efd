// lab_common.h -- a bench bed for ONE family of column kernels outside the library: synthetic columns on the library's field layout,
// hipEvent timing, bitwise comparison of two variants' outputs.  Built by tools/lab/build.sh in seconds (the library takes five
// minutes), run on the GPU box as a plain executable (no Python): many variants per gpurun call.  Not a product path.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define HC(x)                                                                                  \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) {                                                                    \
      std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));   \
      std::exit(2);                                                                            \
    }                                                                                          \
  } while (0)

namespace lab {

struct Rng {   // splitmix64
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed) {}
  uint64_t next() {
    uint64_t z = (s += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
  }
  double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }   // [0, 1)
  double sym() { return 2. * uni() - 1.; }
};

struct DevArr {
  double *d = nullptr;
  size_t n = 0;
  std::vector<double> h;
  void alloc(size_t n_) {
    n = n_;
    h.assign(n, 0.);
    HC(hipMalloc(&d, n * sizeof(double)));
  }
  void up() { HC(hipMemcpy(d, h.data(), n * sizeof(double), hipMemcpyHostToDevice)); }
  void down() { HC(hipMemcpy(h.data(), d, n * sizeof(double), hipMemcpyDeviceToHost)); }
  void zero() { HC(hipMemset(d, 0, n * sizeof(double))); }
  std::vector<double> get() {
    std::vector<double> r(n);
    HC(hipMemcpy(r.data(), d, n * sizeof(double), hipMemcpyDeviceToHost));
    return r;
  }
};

inline size_t count_diff(const std::vector<double> &a, const std::vector<double> &b, double *maxrel = nullptr) {
  size_t nd = 0;
  double mr = 0.;
  for (size_t i = 0; i < a.size(); i++) {
    uint64_t x, y;
    std::memcpy(&x, &a[i], 8);
    std::memcpy(&y, &b[i], 8);
    if (x != y) {
      nd++;
      const double den = std::fabs(a[i]) > 1e-300 ? std::fabs(a[i]) : 1.;
      const double r = std::fabs(a[i] - b[i]) / den;
      if (r > mr || r != r) mr = r;
    }
  }
  if (maxrel) *maxrel = mr;
  return nd;
}

// time `fn` (which launches on stream 0): `reps` launches, hipEvents around each; returns {min, mean} in ms
template <class Fn, class Reset>
inline void time_it(const char *label, int reps, Reset reset, Fn fn, double bytes_alg = 0.) {
  if (const char *only = std::getenv("LAB_ONLY"))   // run only the variants whose label contains this text (for rocprofv3 passes)
    if (!std::strstr(label, only)) return;
  hipEvent_t e0, e1;
  HC(hipEventCreate(&e0));
  HC(hipEventCreate(&e1));
  for (int w = 0; w < 2; w++) {
    reset();
    fn();
  }
  HC(hipDeviceSynchronize());
  double mn = 1e30, sum = 0.;
  for (int r = 0; r < reps; r++) {
    reset();
    HC(hipDeviceSynchronize());
    HC(hipEventRecord(e0, 0));
    fn();
    HC(hipEventRecord(e1, 0));
    HC(hipEventSynchronize(e1));
    float ms = 0.f;
    HC(hipEventElapsedTime(&ms, e0, e1));
    mn = ms < mn ? ms : mn;
    sum += ms;
  }
  std::printf("%-58s min %.4f ms  mean %.4f ms", label, mn, sum / reps);
  if (bytes_alg > 0.) std::printf("  frac %.3f", bytes_alg / (mn * 1e-3) / 8e12);
  std::printf("\n");
  std::fflush(stdout);
  HC(hipEventDestroy(e0));
  HC(hipEventDestroy(e1));
}

}  // namespace lab
