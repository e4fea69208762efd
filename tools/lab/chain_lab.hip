// chain_lab.hip -- what a DEPENDENT chain of f64 operations costs on gfx950: cycles per operation for one wavefront alone on its SIMD,
// with 16 or 64 active lanes, and for the w system's elimination step as nh_fast.h w_column writes it.
#include "lab_common.h"

__device__ __forceinline__ double rcp_rn(double b) {
  double y = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-b, y, 1.0);
  return __builtin_fma(y, e, y);
}
__device__ __forceinline__ double div_rn(double a, double b, double y) {
  const double q0 = a * y;
  const double r = __builtin_fma(-b, q0, a);
  return __builtin_fma(r, y, q0);
}

// mode 0: n dependent fma; 1: n dependent mul; 2: n dependent rcp; 3: n steps of the w elimination (11-op chain + the y chain)
// 4: the same with two independent columns per lane
__global__ void __launch_bounds__(64) chain(int mode, int n, int lanes, double *out, long long *cyc) {
  if ((int)threadIdx.x >= lanes) return;
  double x = 1.0 + 1e-3 * threadIdx.x, a = 0.999, b = 1e-4;
  const long long t0 = clock64();
  if (mode == 0) {
    for (int i = 0; i < n; i++) x = __builtin_fma(x, a, b);
  } else if (mode == 1) {
    for (int i = 0; i < n; i++) x = x * a;
  } else if (mode == 2) {
    for (int i = 0; i < n; i++) x = __builtin_amdgcn_rcp(x) + 0.5;
  } else if (mode == 3) {
    double bet = 2.0 + x, rbet = rcp_rn(bet), y = 0.1;
    const double aa = -50.0 - x, low = -51.0, dm = 0.7, rhs = 0.3;
#pragma unroll 8
    for (int i = 0; i < n; i++) {
      const double gam = div_rn(aa, bet, rbet);
      bet = dm - (aa + low + aa * gam);
      rbet = rcp_rn(bet);
      y = div_rn(rhs - aa * y, bet, rbet);
    }
    x = bet + y;
  } else if (mode == 4) {
    double bet = 2.0 + x, rbet = rcp_rn(bet), y = 0.1;
    double bet2 = 2.5 + x, rbet2 = rcp_rn(bet2), y2 = 0.2;
    const double aa = -50.0 - x, low = -51.0, dm = 0.7, rhs = 0.3;
#pragma unroll 8
    for (int i = 0; i < n; i++) {
      const double gam = div_rn(aa, bet, rbet);
      const double gam2 = div_rn(aa, bet2, rbet2);
      bet = dm - (aa + low + aa * gam);
      bet2 = dm - (aa + low + aa * gam2);
      rbet = rcp_rn(bet);
      rbet2 = rcp_rn(bet2);
      y = div_rn(rhs - aa * y, bet, rbet);
      y2 = div_rn(rhs - aa * y2, bet2, rbet2);
    }
    x = bet + y + bet2 + y2;
  } else if (mode == 5) {   // the forward y chain alone: 5 dependent operations a level
    double y = 0.1;
    const double aa = -50.0 - x, bet = 101.7, rbet = rcp_rn(bet), rhs = 0.3;
#pragma unroll 8
    for (int i = 0; i < n; i++) y = div_rn(rhs - aa * y, bet, rbet);
    x = y;
  } else if (mode == 6) {   // the bet chain alone
    double bet = 2.0 + x, rbet = rcp_rn(bet);
    const double aa = -50.0 - x, low = -51.0, dm = 0.7;
#pragma unroll 8
    for (int i = 0; i < n; i++) {
      const double gam = div_rn(aa, bet, rbet);
      bet = dm - (aa + low + aa * gam);
      rbet = rcp_rn(bet);
    }
    x = bet;
  }
  const long long t1 = clock64();
  out[blockIdx.x * 64 + threadIdx.x] = x;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  double *out;
  long long *cyc;
  const int nb = 4096;
  HC(hipMalloc(&out, nb * 64 * 8));
  HC(hipMalloc(&cyc, nb * 8));
  const char *names[] = {"fma", "mul", "rcp + add", "w elimination step (bet + y)", "two columns per lane", "y chain", "bet chain"};
  for (int mode = 0; mode < 7; mode++)
    for (int lanes : {16, 64})
      for (int blocks : {1, 256, 1024, 2048}) {
        const int n = 4096;
        hipEvent_t e0, e1;
        HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
        chain<<<blocks, 64>>>(mode, n, lanes, out, cyc);
        HC(hipDeviceSynchronize());
        HC(hipEventRecord(e0, 0));
        chain<<<blocks, 64>>>(mode, n, lanes, out, cyc);
        HC(hipEventRecord(e1, 0));
        HC(hipEventSynchronize(e1));
        float ms;
        HC(hipEventElapsedTime(&ms, e0, e1));
        long long c0;
        HC(hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost));
        std::printf("%-32s lanes %2d blocks %4d: %8.2f ns per step (wall), %7.1f clock64 ticks per step\n", names[mode], lanes, blocks,
                    ms * 1e6 / n, (double)c0 / n);
      }
  return 0;
}
