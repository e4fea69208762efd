// op_lab.hip -- issue cost and dependent latency of the f64 operations the column kernels are made of (gfx950), one wavefront alone on a
// SIMD: cycles per operation in a fully dependent chain (latency) and in 8 independent chains (throughput).
#include "lab_common.h"

template <int OP>
__device__ __forceinline__ double op(double x, double a, double b) {
  if (OP == 0) return __builtin_fma(x, a, b);
  if (OP == 1) return x * a;
  if (OP == 2) return x + b;
  if (OP == 3) return __builtin_amdgcn_rcp(x);
  if (OP == 4) return x > a ? x : b;                       // v_cmp + 2 v_cndmask
  if (OP == 5) return __builtin_fmax(x, b);                // v_max_f64
  if (OP == 6) {                                           // DPP row_shr:1 of a double (2 v_mov_dpp)
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x111, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x111, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
  }
  if (OP == 7) return (double)(float)x;                    // cvt pair
  if (OP == 8) return __builtin_amdgcn_ldexp(x, 1);
  return x;
}

template <int OP, int ILP>
__global__ void __launch_bounds__(64) k(int n, double *out, long long *cyc, double a, double b) {
  double x[ILP];
  for (int i = 0; i < ILP; i++) x[i] = 1.0 + 1e-3 * (threadIdx.x + i);
  const long long t0 = clock64();
  for (int it = 0; it < n; it++) {
#pragma unroll
    for (int u = 0; u < 32; u++)
#pragma unroll
      for (int i = 0; i < ILP; i++) x[i] = op<OP>(x[i], a, b);
  }
  const long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < ILP; i++) s += x[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP, int ILP>
void run(const char *name, double *out, long long *cyc) {
  const int n = 64;
  k<OP, ILP><<<1, 64>>>(n, out, cyc, 0.999, 1e-4);
  HC(hipDeviceSynchronize());
  k<OP, ILP><<<1, 64>>>(n, out, cyc, 0.999, 1e-4);
  HC(hipDeviceSynchronize());
  long long c;
  HC(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  std::printf("%-28s %d chain%s: %6.2f cycles per operation\n", name, ILP, ILP > 1 ? "s" : " ", (double)c / (n * 32.0 * ILP));
}

int main() {
  double *out;
  long long *cyc;
  HC(hipMalloc(&out, 64 * 8));
  HC(hipMalloc(&cyc, 8));
#define BOTH(OP, NAME) run<OP, 1>(NAME, out, cyc); run<OP, 2>(NAME, out, cyc); run<OP, 8>(NAME, out, cyc);
  BOTH(0, "v_fma_f64")
  BOTH(1, "v_mul_f64")
  BOTH(2, "v_add_f64")
  BOTH(3, "v_rcp_f64")
  BOTH(4, "cmp + 2 cndmask")
  BOTH(5, "v_max_f64")
  BOTH(6, "dpp row_shr of a double")
  BOTH(7, "cvt f64 -> f32 -> f64")
  BOTH(8, "v_ldexp_f64")
  return 0;
}
