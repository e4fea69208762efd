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
template <class K>
static void go_pool(const K &k, size_t lds_doubles) {
  int rc = launch_2w(Dim3{(unsigned)k.pool, 1, 1}, lds_doubles, 0, k);
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

int main(int argc, char **argv) {
  const int nx = argc > 1 ? std::atoi(argv[1]) : 384, km = argc > 2 ? std::atoi(argv[2]) : 127;
  const int reps = argc > 3 ? std::atoi(argv[3]) : 10;
  State s;
  make_state(s, nx, km);
  const double cells = (double)nx * nx * km;
  std::printf("riem_lab %d x %d x %d\n", nx, nx, km);
  const size_t lds_old = (size_t)kFNBuf * kFBuf;
  auto rs = [&] { reset(s); };
  // ---- the library's kernels as they are: the reference for everything below
  Out3 ref3, ref3l;
  OutC refc;
  {
    auto k = make3<RiemFast<false, true>>(s, 22.5, 0);
    auto kl = make3<RiemFast<false, true>>(s, 22.5, 1);
    reset(s); go(kl, lds_old); HC(hipDeviceSynchronize());   // (pe, pk, peln are written by the last call only: fill them first)
    reset(s); go(k, lds_old); ref3 = grab3(s);
    lab::time_it("riem3 RiemFast (library)", reps, rs, [&] { go(k, lds_old); }, cells * 72);
    reset(s); go(kl, lds_old); ref3l = grab3(s);
    auto kc = makec<RiemFast<true, true>>(s, 11.25);
    reset(s); go(kc, lds_old); refc = grabc(s);
    lab::time_it("riemC RiemFast (library)", reps, rs, [&] { go(kc, lds_old); }, cells * 48);
    for (int pr : {2, 7}) {
      auto kp = k; kp.probe = pr;
      auto kcp = kc; kcp.probe = pr;
      char lb[96];
      std::snprintf(lb, sizeof lb, "riem3 RiemFast probe %d (wrong results)", pr);
      lab::time_it(lb, reps, rs, [&] { go(kp, lds_old); }, cells * 72);
      std::snprintf(lb, sizeof lb, "riemC RiemFast probe %d (wrong results)", pr);
      lab::time_it(lb, reps, rs, [&] { go(kcp, lds_old); }, cells * 48);
    }
    for (int opt : {3, 7}) {
      auto ko = k; ko.opt = opt;
      auto kco = kc; kco.opt = opt;
      char lb[96];
      std::snprintf(lb, sizeof lb, "riem3 RiemFast opt %d", opt);
      lab::time_it(lb, reps, rs, [&] { go(ko, lds_old + 8); }, cells * 72);
      std::snprintf(lb, sizeof lb, "riemC RiemFast opt %d", opt);
      lab::time_it(lb, reps, rs, [&] { go(kco, lds_old + 8); }, cells * 48);
      reset(s); go(ko, lds_old + 8); same3(lb, ref3, grab3(s));
      reset(s); go(kco, lds_old + 8); samec(lb, refc, grabc(s));
      if (opt == 7) {
        ko.stg_mode = 1; ko.stg_first = 512; ko.stg_ticks = 2500;
        kco.stg_mode = 1; kco.stg_first = 512; kco.stg_ticks = 2500;
        lab::time_it("riem3 RiemFast opt 7 + stagger 25 us", reps, rs, [&] { go(ko, lds_old + 8); }, cells * 72);
        lab::time_it("riemC RiemFast opt 7 + stagger 25 us", reps, rs, [&] { go(kco, lds_old + 8); }, cells * 48);
      }
    }
    for (int pool : {512, 1024})
      for (int opt : {0, 3}) {
        auto ko = k; ko.opt = opt; ko.pool = pool;
        auto kco = kc; kco.opt = opt; kco.pool = pool;
        char lb[96];
        std::snprintf(lb, sizeof lb, "riem3 RiemFast pool %d opt %d", pool, opt);
        lab::time_it(lb, reps, rs, [&] { go_pool(ko, lds_old + 8); }, cells * 72);
        std::snprintf(lb, sizeof lb, "riemC RiemFast pool %d opt %d", pool, opt);
        lab::time_it(lb, reps, rs, [&] { go_pool(kco, lds_old + 8); }, cells * 48);
        reset(s); go_pool(ko, lds_old + 8); same3(lb, ref3, grab3(s));
        reset(s); go_pool(kco, lds_old + 8); samec(lb, refc, grabc(s));
        ko.stg_mode = 1; ko.stg_first = 512; ko.stg_ticks = 2500;
        kco.stg_mode = 1; kco.stg_first = 512; kco.stg_ticks = 2500;
        std::snprintf(lb, sizeof lb, "riem3 RiemFast pool %d opt %d + stagger 25 us", pool, opt);
        lab::time_it(lb, reps, rs, [&] { go_pool(ko, lds_old + 8); }, cells * 72);
        std::snprintf(lb, sizeof lb, "riemC RiemFast pool %d opt %d + stagger 25 us", pool, opt);
        lab::time_it(lb, reps, rs, [&] { go_pool(kco, lds_old + 8); }, cells * 48);
      }
    {
      auto kp = k; kp.probe = 2; kp.pool = 512; kp.opt = 3;
      lab::time_it("riem3 RiemFast pool 512 probe 2 (wrong results)", reps, rs, [&] { go_pool(kp, lds_old + 8); }, cells * 72);
      kp.probe = 7;
      lab::time_it("riem3 RiemFast pool 512 probe 7 (wrong results)", reps, rs, [&] { go_pool(kp, lds_old + 8); }, cells * 72);
    }
    // staggered start
    for (int mode : {1})
      for (int ticks : {2500}) {
        auto ks = k; ks.stg_mode = mode; ks.stg_first = 512; ks.stg_ticks = ticks;
        auto kcs = kc; kcs.stg_mode = mode; kcs.stg_first = 512; kcs.stg_ticks = ticks;
        char lb[96];
        std::snprintf(lb, sizeof lb, "riem3 RiemFast stagger mode %d, %d us", mode, ticks / 100);
        lab::time_it(lb, reps, rs, [&] { go(ks, lds_old); }, cells * 72);
        std::snprintf(lb, sizeof lb, "riemC RiemFast stagger mode %d, %d us", mode, ticks / 100);
        lab::time_it(lb, reps, rs, [&] { go(kcs, lds_old); }, cells * 48);
        if (mode == 1 && ticks == 5000) {
          reset(s); go(ks, lds_old); same3("riem3 staggered vs library", ref3, grab3(s));
          reset(s); go(kcs, lds_old); samec("riemC staggered vs library", refc, grabc(s));
        }
      }
    for (int pr : {2}) {
      auto kp = k; kp.probe = pr; kp.stg_mode = 1; kp.stg_first = 512; kp.stg_ticks = 5000;
      lab::time_it("riem3 RiemFast probe 2 + stagger 50 us", reps, rs, [&] { go(kp, lds_old); }, cells * 72);
    }
  }
#ifdef LAB_HAVE_NEW
  lab_new(s, ref3, ref3l, refc, reps, cells);
#endif
  return 0;
}
