#include <hip/hip_runtime.h>
#include <cstdio>
template <int N> struct Big { double v[N]; const double *p[8]; };
template <int N> struct Grp { Big<N> f[6]; };
template <int N> __global__ void k(const Grp<N> g, double *out) {
  const int face = blockIdx.z;
  const Big<N> &f = g.f[face];
  double s = 0;
  for (int i = threadIdx.x; i < N; i += blockDim.x) s += f.v[i];
  atomicAdd(out + face, s);
}
template <int N> int run() {
  Grp<N> g;
  for (int f = 0; f < 6; f++) for (int i = 0; i < N; i++) g.f[f].v[i] = f + 1;
  double *out; hipMalloc(&out, 6 * 8); hipMemset(out, 0, 48);
  hipLaunchKernelGGL(k<N>, dim3(1, 1, 6), dim3(64), 0, 0, g, out);
  hipError_t e = hipGetLastError();
  hipError_t e2 = hipDeviceSynchronize();
  double h[6]; hipMemcpy(h, out, 48, hipMemcpyDeviceToHost);
  printf("sizeof %zu: launch %s sync %s  out0 %g out5 %g (expect %d %d)\n", sizeof(g), hipGetErrorString(e), hipGetErrorString(e2), h[0], h[5], N, 6 * N);
  return 0;
}
int main() { run<64>(); run<100>(); run<160>(); run<330>(); run<1300>(); return 0; }
