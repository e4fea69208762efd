// buffer-instruction semantics probe (gfx950): which offsets take part in the raw-buffer range check?
//   hipcc --offload-arch=gfx950 -O2 tools/probe/buf_oob.hip -o variants/buf_oob && variants/buf_oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned u2;
__global__ void k(double *p, double *out, int num_records, unsigned voff_bad, int soff) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p, 0, num_records, 0x00020000);
  const unsigned lane = threadIdx.x;
  // lanes 0..31 in range, lanes 32..63 get the "bad" offset
  const unsigned vo = lane < 32 ? lane * 8 : voff_bad;
  const double x = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, vo, soff, 0));
  out[lane] = x;
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, 1000.0 + lane), r, vo, soff, 0);
}
int main() {
  const int n = 4096;
  std::vector<double> h(n);
  double *d, *o;
  hipMalloc(&d, n * 8); hipMalloc(&o, 64 * 8);
  struct { int nr; unsigned bad; int soff; const char *what; } cases[] = {
    {512, 0x80000000u, 0, "num_records 512, bad voffset 0x80000000, soffset 0"},
    {0x7fffffff, 0x80000000u, 0, "num_records 0x7fffffff, bad voffset 0x80000000, soffset 0"},
    {512, 0x80000000u, 8192, "num_records 512, soffset 8192 (beyond num_records): is soffset range-checked?"},
    {512, 600u, 0, "num_records 512, bad voffset 600 (just beyond)"},
    {512, 504u, 8192, "num_records 512, voffset 504 (last element), soffset 8192"},
  };
  for (auto &c : cases) {
    for (int i = 0; i < n; i++) h[i] = i;
    hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o, c.nr, c.bad, c.soff);
    std::vector<double> ho(64), hd(n);
    hipMemcpy(ho.data(), o, 64 * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hd.data(), d, n * 8, hipMemcpyDeviceToHost);
    int changed = 0, first = -1, last = -1;
    for (int i = 0; i < n; i++) if (hd[i] != i) { changed++; if (first < 0) first = i; last = i; }
    printf("%s\n  loads: lane0 %.0f lane31 %.0f lane32 %.0f lane63 %.0f ; stores changed %d elements [%d..%d] (d[%d]=%.0f)\n", c.what, ho[0], ho[31], ho[32],
           ho[63], changed, first, last, first < 0 ? 0 : first, first < 0 ? 0. : hd[first]);
  }
  return 0;
}
