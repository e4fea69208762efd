// riem_lab.hip -- the Riemann solvers' levels-across-the-lanes kernels (csrc/nh_fast.h) on a synthetic C384 L127 tile, outside the
// library: variants timed against each other and compared bit for bit.  tools/lab/build.sh riem_lab; run on the GPU box.
//   riem_lab [nx] [km]
#include "lab_common.h"

#include "../../gfdl_atmos_cubed_sphere_amd/csrc/fv3_launch.h"
#include "../../gfdl_atmos_cubed_sphere_amd/csrc/nh_fast.h"
#ifdef LAB_HAVE_NEW
#include "../../gfdl_atmos_cubed_sphere_amd/csrc/nh_col.h"
#endif

using namespace fv3;
using lab::DevArr;

struct State {
  Grid g{};
  int km;
  NhConsts cn{};
  DevArr zs, hs, pt, delp, ws_cc, ws_a, w, zh, gz, delz, ppe, pk3, pef, pe, pk, peln, w0, zh0;
};

static void make_state(State &s, int nx, int km) {
  Grid &g = s.g;
  std::memset(&g, 0, sizeof(g));
  g.is = 1; g.ie = nx; g.js = 1; g.je = nx;
  g.isd = 1 - NG; g.ied = nx + NG; g.jsd = 1 - NG; g.jed = nx + NG;
  g.npx = nx + 1; g.npy = nx + 1; g.npz = km;
  g.nid = nx + 2 * NG; g.njd = nx + 2 * NG; g.nx = nx; g.ny = nx;
  g.grid_type = 4;
  s.km = km;
  const double GRAV = 9.80665, RDGAS = 287.05, CP = 1004.6, KAPPA = RDGAS / CP, PTOP = 300.;
  s.cn.grav = GRAV; s.cn.rdgas = RDGAS; s.cn.cp_air = CP; s.cn.akap = KAPPA; s.cn.ptop = PTOP; s.cn.p_fac = 0.05; s.cn.a_imp = 1.0;
  const size_t nA = g.nA(), nCC = g.nCC();
  s.zs.alloc(nA); s.hs.alloc(nA); s.pt.alloc(nA * km); s.delp.alloc(nA * km); s.ws_cc.alloc(nCC); s.ws_a.alloc(nA);
  s.w.alloc(nA * km); s.zh.alloc(nA * (km + 1)); s.gz.alloc(nA * (km + 1)); s.delz.alloc(nCC * km);
  s.ppe.alloc(nA * (km + 1)); s.pk3.alloc(nA * (km + 1)); s.pef.alloc(nA * (km + 1));
  s.pe.alloc((size_t)(nx + 2) * (nx + 2) * (km + 1)); s.pk.alloc(nCC * (km + 1)); s.peln.alloc(nCC * (km + 1));
  s.w0.alloc(nA * km); s.zh0.alloc(nA * (km + 1));
  lab::Rng r(11);
  std::vector<double> sig(km + 1);
  for (int k = 0; k <= km; k++) sig[k] = std::pow((double)k / km, 1.5);
  for (size_t c = 0; c < nA; c++) {
    const double ps = 1.0e5 * (1. + 0.01 * r.sym());
    s.zs.h[c] = 50. * r.uni();
    s.hs.h[c] = s.zs.h[c] * GRAV;
    s.ws_a.h[c] = 0.1 * r.sym();
    std::vector<double> pe(km + 1), dz(km);
    for (int k = 0; k <= km; k++) pe[k] = PTOP + (ps - PTOP) * sig[k];
    for (int k = 0; k < km; k++) {
      const double dp = pe[k + 1] - pe[k], pm = dp / std::log(pe[k + 1] / pe[k]);
      const double T = 300. - 60. * (1. - sig[k + 1]) + 2. * r.sym();
      const double pt = T * std::pow(pm, -KAPPA);
      s.delp.h[(size_t)k * nA + c] = dp;
      s.pt.h[(size_t)k * nA + c] = pt;
      dz[k] = -dp / GRAV * RDGAS * pt * std::pow(pm, KAPPA - 1.) * (1. + 0.02 * r.sym());
      s.w0.h[(size_t)k * nA + c] = 0.5 * r.sym();
    }
    s.zh0.h[(size_t)km * nA + c] = s.zs.h[c];
    for (int k = km - 1; k >= 0; k--) s.zh0.h[(size_t)k * nA + c] = s.zh0.h[(size_t)(k + 1) * nA + c] - dz[k];
  }
  for (size_t c = 0; c < nCC; c++) s.ws_cc.h[c] = 0.1 * r.sym();
  s.zs.up(); s.hs.up(); s.pt.up(); s.delp.up(); s.ws_cc.up(); s.ws_a.up(); s.w0.up(); s.zh0.up();
}

static void reset(State &s) {
  HC(hipMemcpyAsync(s.w.d, s.w0.d, s.w.n * 8, hipMemcpyDeviceToDevice, 0));
  HC(hipMemcpyAsync(s.zh.d, s.zh0.d, s.zh.n * 8, hipMemcpyDeviceToDevice, 0));
  HC(hipMemcpyAsync(s.gz.d, s.zh0.d, s.gz.n * 8, hipMemcpyDeviceToDevice, 0));
}

struct Out3 { std::vector<double> zh, w, delz, ppe, pk3, pe, pk, peln; };
struct OutC { std::vector<double> gz, pef; };

template <class K>
static K make3(State &s, double dt, int last_call) {
  return K{s.g, s.km, dt, s.cn, s.zs.d, s.pt.d, s.delp.d, s.ws_cc.d, s.w.d, s.zh.d, s.delz.d, s.ppe.d, s.pk3.d, s.pe.d, s.pk.d, s.peln.d, nullptr,
           0, last_call, 0};
}
template <class K>
static K makec(State &s, double dt) {
  return K{s.g, s.km, dt, s.cn, s.hs.d, s.pt.d, s.delp.d, s.ws_a.d, s.w.d, s.gz.d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, s.pef.d, 0, 0, 0};
}
template <class K>
static void go(const K &k, size_t lds_doubles) {
  int rc = launch_2w(Dim3{(unsigned)k.nblocks_x(), (unsigned)k.nrows(), 1}, lds_doubles, 0, k);
  if (rc) { std::fprintf(stderr, "launch failed %d\n", rc); std::exit(3); }
}
static Out3 grab3(State &s) {
  HC(hipDeviceSynchronize());
  return Out3{s.zh.get(), s.w.get(), s.delz.get(), s.ppe.get(), s.pk3.get(), s.pe.get(), s.pk.get(), s.peln.get()};
}
static OutC grabc(State &s) {
  HC(hipDeviceSynchronize());
  return OutC{s.gz.get(), s.pef.get()};
}
static bool same3(const char *what, const Out3 &a, const Out3 &b) {
  size_t n = 0;
  n += lab::count_diff(a.zh, b.zh); n += lab::count_diff(a.w, b.w); n += lab::count_diff(a.delz, b.delz); n += lab::count_diff(a.ppe, b.ppe);
  n += lab::count_diff(a.pk3, b.pk3); n += lab::count_diff(a.pe, b.pe); n += lab::count_diff(a.pk, b.pk); n += lab::count_diff(a.peln, b.peln);
  std::printf("  %-40s %s (%zu words differ)\n", what, n ? "DIFFERENT" : "bit-identical", n);
  return n == 0;
}
static bool samec(const char *what, const OutC &a, const OutC &b) {
  size_t n = lab::count_diff(a.gz, b.gz) + lab::count_diff(a.pef, b.pef);
  std::printf("  %-40s %s (%zu words differ)\n", what, n ? "DIFFERENT" : "bit-identical", n);
  return n == 0;
}

void pass_bench(State &s);
void fwd_bench(int km);
int main(int argc, char **argv) {
  const int nx = argc > 1 ? std::atoi(argv[1]) : 384, km = argc > 2 ? std::atoi(argv[2]) : 127;
  const int reps = argc > 3 ? std::atoi(argv[3]) : 10;
  State s;
  make_state(s, nx, km);
  const double cells = (double)nx * nx * km;
  std::printf("riem_lab %d x %d x %d\n", nx, nx, km);
  auto rs = [&] { reset(s); };
  // ---- opt = 0: the kernels as round 5 left them: the reference for everything below
  Out3 ref3, ref3l;
  OutC refc;
  const size_t lds_n = RiemFast<false, true>::kLdsDoubles;
  {
    auto k = make3<RiemFast<false, true>>(s, 22.5, 0);
    auto kl = make3<RiemFast<false, true>>(s, 22.5, 1);
    auto kc = makec<RiemFast<true, true>>(s, 11.25);
    k.opt = kl.opt = kc.opt = 0;
    reset(s); go(kl, lds_n); HC(hipDeviceSynchronize());   // (pe, pk, peln are written by the last call only: fill them first)
    reset(s); go(k, lds_n); ref3 = grab3(s);
    lab::time_it("riem3 RiemFast opt 0 (round 5)", reps, rs, [&] { go(k, lds_n); }, cells * 72);
    reset(s); go(kl, lds_n); ref3l = grab3(s);
    reset(s); go(kc, lds_n); refc = grabc(s);
    lab::time_it("riemC RiemFast opt 0 (round 5)", reps, rs, [&] { go(kc, lds_n); }, cells * 48);
    for (int pr : {2, 7}) {
      auto kp = k; kp.probe = pr;
      auto kcp = kc; kcp.probe = pr;
      char lb[96];
      std::snprintf(lb, sizeof lb, "riem3 RiemFast opt 0 probe %d (wrong results)", pr);
      lab::time_it(lb, reps, rs, [&] { go(kp, lds_n); }, cells * 72);
      std::snprintf(lb, sizeof lb, "riemC RiemFast opt 0 probe %d (wrong results)", pr);
      lab::time_it(lb, reps, rs, [&] { go(kcp, lds_n); }, cells * 48);
    }
    for (int opt : {1, 2, 4, 7}) {
      auto ko = k; ko.opt = opt;
      auto klo = kl; klo.opt = opt;
      auto kco = kc; kco.opt = opt;
      char lb[96];
      std::snprintf(lb, sizeof lb, "riem3 RiemFast opt %d", opt);
      lab::time_it(lb, reps, rs, [&] { go(ko, lds_n); }, cells * 72);
      reset(s); go(ko, lds_n); same3(lb, ref3, grab3(s));
      std::snprintf(lb, sizeof lb, "riem3 RiemFast opt %d, last call", opt);
      reset(s); go(klo, lds_n); same3(lb, ref3l, grab3(s));
      std::snprintf(lb, sizeof lb, "riemC RiemFast opt %d", opt);
      lab::time_it(lb, reps, rs, [&] { go(kco, lds_n); }, cells * 48);
      reset(s); go(kco, lds_n); samec(lb, refc, grabc(s));
    }
  }
#ifdef FV3_LAB_TRACE
  pass_bench(s);
  fwd_bench(km);
  {
    long long *tr;
    HC(hipMalloc(&tr, 4 * 16 * 8));
    for (int cg = 0; cg < 4; cg++)
      for (int blk : {100, 4000}) {
        const size_t lds_tr = cg >= 2 ? (size_t)12800 : lds_n;   // 100 KB: one workgroup per CU
        HC(hipMemset(tr, 0, 4 * 16 * 8));
        reset(s);
        if (cg & 1) { auto kc = makec<RiemFast<true, true>>(s, 11.25); kc.trace = tr; kc.trace_blk = blk; go(kc, lds_tr); }
        else { auto k = make3<RiemFast<false, true>>(s, 22.5, 0); k.trace = tr; k.trace_blk = blk; go(k, lds_tr); }
        HC(hipDeviceSynchronize());
        long long h[64];
        HC(hipMemcpy(h, tr, sizeof h, hipMemcpyDeviceToHost));
        std::printf("%strace %s block %d: cycles since the block's start (a = the wavefront arrives at the barrier, l = leaves)\n",
                    cg >= 2 ? "ONE workgroup per CU: " : "", (cg & 1) ? "riemC" : "riem3", blk);
        const char *nm[] = {"staged a", "l", "first half a", "l", "pass a", "l", "second half a", "l"};
        for (int w = 0; w < 4; w++) {
          const long long t0 = h[w * 16 + 15];
          std::printf("  wave %d:", w);
          for (int n = 0; n < 8; n++) std::printf(" %s %lld", nm[n], h[w * 16 + n] - t0);
          std::printf("\n");
        }
      }
  }
#endif
#ifdef LAB_HAVE_NEW
  lab_new(s, ref3, ref3l, refc, reps, cells);
#endif
  return 0;
}

// ---- the w pass alone: LDS filled with regular coefficients, one workgroup, clock64 around the sweep --------------------------------
template <int WHICH>
__global__ void __launch_bounds__(256) pass_only(RiemFast<false, true> k, long long *cyc, double *out) {
  extern __shared__ double fv3_lds[];
  double *B0 = fv3_lds, *B1 = fv3_lds + kFBuf, *B2 = fv3_lds + 2 * kFBuf;
  const int tid = threadIdx.x;
  for (int idx = tid; idx < kFBuf; idx += 256) {
    B0[idx] = -50. - 1e-3 * (idx % 97);
    B1[idx] = 0.7 + 1e-4 * (idx % 31);
    B2[idx] = 0.3 + 1e-3 * (idx % 13);
  }
  __syncthreads();
  const long long t0 = clock64();
  if ((tid >> 6) == 0 && (tid & 63) < kFC) {
    const int col = tid & 63;
    if (WHICH == 0) k.w_column(B0 + col * kFP, B1 + col * kFP, B2 + col * kFP);
    else k.w_column2(B0 + col * kFP, B1 + col * kFP, B2 + col * kFP);
  }
  const long long t1 = clock64();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  __syncthreads();
  out[blockIdx.x * 256 + tid] = B2[tid] + B0[tid];
}
void pass_bench(State &s) {
  long long *cyc;
  double *out;
  HC(hipMalloc(&cyc, 8 * 1024));
  HC(hipMalloc(&out, 8 * 256 * 1024));
  auto k = make3<RiemFast<false, true>>(s, 22.5, 0);
  HC(hipFuncSetAttribute(reinterpret_cast<const void *>(&pass_only<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  HC(hipFuncSetAttribute(reinterpret_cast<const void *>(&pass_only<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  for (int which = 0; which < 2; which++)
    for (int nb : {1, 256}) {
      if (which) pass_only<1><<<nb, 256, 100 * 1024>>>(k, cyc, out);
      else pass_only<0><<<nb, 256, 100 * 1024>>>(k, cyc, out);
      HC(hipDeviceSynchronize());
      long long c;
      HC(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
      std::printf("pass alone (%s), %d workgroups: %lld cycles = %.1f per level\n", which ? "w_column2" : "w_column", nb, c, (double)c / s.km);
    }
}

// ---- attribution of the forward sweep's time: MODE 0 as w_column2; 1 no LDS stores; 2 no LDS loads inside the loop; 3 neither ----
template <int MODE>
__global__ void __launch_bounds__(256) fwd_only(long long *cyc, double *out, int km) {
  extern __shared__ double fv3_lds[];
  double *B0 = fv3_lds, *B1 = fv3_lds + kFBuf, *B2 = fv3_lds + 2 * kFBuf;
  const int tid = threadIdx.x;
  for (int idx = tid; idx < kFBuf; idx += 256) {
    B0[idx] = -50. - 1e-3 * (idx % 97);
    B1[idx] = 0.7 + 1e-4 * (idx % 31);
    B2[idx] = 0.3 + 1e-3 * (idx % 13);
  }
  __syncthreads();
  const long long t0 = clock64();
  double acc = 0.;
  if ((tid >> 6) == 0 && (tid & 63) < kFC) {
    const int col = tid & 63;
    double *A = B0 + col * kFP, *D = B1 + col * kFP, *R = B2 + col * kFP;
    double bet = 1., rbet = 1., y = 0.;
    const int nch = (km + kFL - 1) / kFL, nch2 = (nch + 1) & ~1;
    double a0[kFL + 1], d0[kFL], r0[kFL], a1[kFL + 1], d1[kFL], r1[kFL];
    for (int u = 0; u < kFL; u++) { a0[u] = A[u]; d0[u] = D[u]; r0[u] = R[u]; }
    a0[kFL] = A[kFS];
    for (int u = 0; u < kFL; u++) { a1[u] = A[kFS + u]; d1[u] = D[kFS + u]; r1[u] = R[kFS + u]; }
    a1[kFL] = A[2 * kFS];
    auto chunk = [&](const double *a, const double *dc, const double *rc, double *gv, double *yv) {
#pragma unroll
      for (int u = 0; u < kFL; u++) {
        const double av = a[u], low = a[u + 1];
        const double gam = div_rn(av, bet, rbet);
        bet = dc[u] - (av + low + av * gam);
        rbet = rcp_rn(bet);
        y = div_rn(rc[u] - av * y, bet, rbet);
        gv[u] = gam;
        yv[u] = y;
      }
    };
    for (int ch = 0; ch < nch2; ch += 2) {
      if (!(MODE & 2)) {
        const int o = (ch + 1) * kFS;
        for (int u = 0; u < kFL; u++) { a1[u] = A[o + u]; d1[u] = D[o + u]; r1[u] = R[o + u]; }
        a1[kFL] = A[o + kFS];
      }
      double gv[kFL], yv[kFL];
      chunk(a0, d0, r0, gv, yv);
      if (!(MODE & 1)) {
        const int o = ch * kFS;
        for (int u = 0; u < kFL; u++) { A[o + u] = gv[u]; R[o + u] = yv[u]; }
      } else {
        for (int u = 0; u < kFL; u++) acc += gv[u] + yv[u];
      }
      if (!(MODE & 2) && ch + 2 < nch2) {
        const int o = (ch + 2) * kFS;
        for (int u = 0; u < kFL; u++) { a0[u] = A[o + u]; d0[u] = D[o + u]; r0[u] = R[o + u]; }
        a0[kFL] = A[o + kFS];
      }
      chunk(a1, d1, r1, gv, yv);
      if (!(MODE & 1)) {
        const int o = (ch + 1) * kFS;
        for (int u = 0; u < kFL; u++) { A[o + u] = gv[u]; R[o + u] = yv[u]; }
      } else {
        for (int u = 0; u < kFL; u++) acc += gv[u] + yv[u];
      }
    }
    acc += bet + y;
  }
  const long long t1 = clock64();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  __syncthreads();
  out[blockIdx.x * 256 + tid] = B2[tid] + B0[tid] + acc;
}
template <int MODE>
static void fwd_one(long long *cyc, double *out, int km) {
  HC(hipFuncSetAttribute(reinterpret_cast<const void *>(&fwd_only<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  fwd_only<MODE><<<1, 256, 100 * 1024>>>(cyc, out, km);
  HC(hipDeviceSynchronize());
  long long c;
  HC(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  std::printf("forward sweep alone, mode %d (1: no LDS stores, 2: no LDS loads in the loop): %lld cycles = %.1f per level\n", MODE, c, (double)c / km);
}
void fwd_bench(int km) {
  long long *cyc;
  double *out;
  HC(hipMalloc(&cyc, 8 * 16));
  HC(hipMalloc(&out, 8 * 256 * 16));
  fwd_one<0>(cyc, out, km);
  fwd_one<1>(cyc, out, km);
  fwd_one<2>(cyc, out, km);
  fwd_one<3>(cyc, out, km);
}
