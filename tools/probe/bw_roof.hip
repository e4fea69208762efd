// achievable HBM bandwidth of streaming kernels with R read and W write streams (fp64 rows, one element per lane and step):
//   hipcc --offload-arch=gfx950 -O3 tools/probe/bw_roof.hip -o variants/bw_roof && variants/bw_roof
// The "roof" the marching kernels can be compared with: what plain load -> add -> store loops reach on this part with the same mix of streams.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int R, int W, bool NT>
__global__ void __launch_bounds__(256) stream(const double *const *in, double *const *out, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i < n; i += stride) {
    double s = 0.;
#pragma unroll
    for (int r = 0; r < R; r++) s += in[r][i];
    if (W == 0 && s == 1.2345e300) out[0][i] = s;   // keeps the loads of the read-only mix alive
#pragma unroll
    for (int w = 0; w < W; w++) {
      if (NT) __builtin_nontemporal_store(s + w, out[w] + i);
      else out[w][i] = s + w;
    }
  }
}
template <int R, int W, bool NT>
double run(const double *const *din, double *const *dout, size_t n, int blocks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; i++) stream<R, W, NT><<<blocks, 256>>>(din, dout, n);
  hipEventRecord(e0);
  const int reps = 10;
  for (int i = 0; i < reps; i++) stream<R, W, NT><<<blocks, 256>>>(din, dout, n);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return (double)(R + W) * n * 8 * reps / (ms * 1e-3) / 1e12;
}
int main() {
  const size_t n = (size_t)390 * 390 * 127;  // one C384L127 field
  const int NA = 16;
  std::vector<double *> hin(NA), hout(NA);
  for (int i = 0; i < NA; i++) { hipMalloc(&hin[i], n * 8); hipMalloc(&hout[i], n * 8); hipMemset(hin[i], 0, n * 8); }
  double **din, **dout;
  hipMalloc(&din, NA * 8); hipMalloc(&dout, NA * 8);
  hipMemcpy(din, hin.data(), NA * 8, hipMemcpyHostToDevice);
  hipMemcpy(dout, hout.data(), NA * 8, hipMemcpyHostToDevice);
  for (int blocks : {2048, 8192, 75451}) {
    printf("blocks %d:  R1W1 %.2f / nt %.2f   R5W10 (c_sw) %.2f / nt %.2f   R12W13 (d_sw transport) %.2f / nt %.2f   R9W3 (momentum) %.2f / nt %.2f   R8W0 %.2f  R0W8 %.2f / nt %.2f TB/s\n", blocks,
           run<1, 1, false>(din, dout, n, blocks), run<1, 1, true>(din, dout, n, blocks), run<5, 10, false>(din, dout, n, blocks), run<5, 10, true>(din, dout, n, blocks),
           run<12, 13, false>(din, dout, n, blocks), run<12, 13, true>(din, dout, n, blocks), run<9, 3, false>(din, dout, n, blocks), run<9, 3, true>(din, dout, n, blocks),
           run<8, 0, false>(din, dout, n, blocks), run<0, 8, false>(din, dout, n, blocks), run<0, 8, true>(din, dout, n, blocks));
  }
  return 0;
}
