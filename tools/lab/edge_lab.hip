// edge_lab.hip -- edge_profile with the levels across the lanes (csrc/nh_fast.h EdgeProfileLds) on a C384 L127 tile, outside the library:
// the wavefront-built row coefficients against the host's table, the two fields of a pair one after the other against side by side;
// timed and compared bit for bit.  tools/lab/build.sh edge_lab; run on the GPU box.
//   edge_lab [nx] [km] [reps]
#ifndef FV3_LAB_EDGE_TABLE_ONLY
#define FV3_LAB_EDGE_OLD_PATH 1   // the rows built by the wavefront (round 5), for the comparison
#endif
#include "lab_common.h"

#include "../../gfdl_atmos_cubed_sphere_amd/csrc/fv3_launch.h"
#include "../../gfdl_atmos_cubed_sphere_amd/csrc/nh_fast.h"

using namespace fv3;
using lab::DevArr;
#ifdef LAB_EDGE_WAVES
namespace fv3 {
template <>
struct tile_waves<EdgeProfileLds> { static constexpr int value = LAB_EDGE_WAVES; };
}
#endif

int main(int argc, char **argv) {
  const int nx = argc > 1 ? std::atoi(argv[1]) : 384, km = argc > 2 ? std::atoi(argv[2]) : 127;
  const int reps = argc > 3 ? std::atoi(argv[3]) : 10;
  Grid g;
  std::memset(&g, 0, sizeof(g));
  g.is = 1; g.ie = nx; g.js = 1; g.je = nx;
  g.isd = 1 - NG; g.ied = nx + NG; g.jsd = 1 - NG; g.jed = nx + NG;
  g.npx = nx + 1; g.npy = nx + 1; g.npz = km;
  g.nid = nx + 2 * NG; g.njd = nx + 2 * NG; g.nx = nx; g.ny = nx;
  g.grid_type = 4;
  // the coefficients as fv3_set_dp_ref computes them (nh_utils.F90:1640-1662)
  std::vector<double> dp0(km), co(4 * km, 0.);
  for (int k = 0; k < km; k++) dp0[k] = 300. * (1. + 0.9 * std::sin(0.11 * k)) + 3. * k;
  double *gk = co.data(), *bet = gk + km, *gam = bet + km, *rbet = gam + km;
  EdgeCoef ec{};
  const double g0 = dp0[1] / dp0[0];
  ec.xt1_top = 2. * g0 * (g0 + 1.);
  ec.bet_top = g0 * (g0 + 0.5);
  gam[0] = (1. + g0 * (g0 + 1.5)) / ec.bet_top;
  double gkk = 0.;
  for (int k = 2; k <= km; k++) {
    gkk = dp0[k - 2] / dp0[k - 1];
    gk[k - 1] = gkk;
    bet[k - 1] = 2. + 2. * gkk - gam[k - 2];
    gam[k - 1] = gkk / bet[k - 1];
    rbet[k - 1] = 1. / bet[k - 1];
  }
  ec.a_bot = 1. + gkk * (gkk + 1.5);
  ec.xt1_bot = 2. * gkk * (gkk + 1.);
  ec.gk_bot = gkk;
  DevArr cod, tab;
  cod.alloc(4 * km);
  cod.h = co;
  cod.up();
  ec.gk = cod.d; ec.bet = cod.d + km; ec.gam = cod.d + 2 * km;
  tab.alloc(EdgeProfileLds::kTabDoubles);
  edge_rows(tab.h.data(), km, gk, bet, gam, rbet, ec.bet_top, ec.a_bot, ec.gk_bot);
  tab.up();

  const size_t nCX = g.nCX(), nCY = g.nCY();
  DevArr crx, xfx, cry, yfx, cxa, xfa, cya, yfa;
  crx.alloc(nCX * km); xfx.alloc(nCX * km); cry.alloc(nCY * km); yfx.alloc(nCY * km);
  cxa.alloc(nCX * (km + 1)); xfa.alloc(nCX * (km + 1)); cya.alloc(nCY * (km + 1)); yfa.alloc(nCY * (km + 1));
  lab::Rng r(5);
  for (auto *a : {&crx, &xfx, &cry, &yfx}) {
    for (size_t i = 0; i < a->n; i++) a->h[i] = 0.3 * r.sym() + 0.1 * std::sin(1e-3 * (double)i);
    a->up();
  }
  std::printf("edge_lab %d x %d x %d\n", nx, nx, km);
  const double bytes = (double)(nCX + nCY) * (2. * km + 2. * (km + 1)) * 8.;
  auto none = [] {};
  auto zero = [&] { cxa.zero(); xfa.zero(); cya.zero(); yfa.zero(); };
  auto grab = [&] {
    HC(hipDeviceSynchronize());
    std::vector<std::vector<double>> o{cxa.get(), xfa.get(), cya.get(), yfa.get()};
    return o;
  };
  auto same = [&](const char *what, const std::vector<std::vector<double>> &a, const std::vector<std::vector<double>> &b) {
    size_t n = 0;
    for (size_t i = 0; i < a.size(); i++) n += lab::count_diff(a[i], b[i]);
    std::printf("  %-40s %s (%zu words differ)\n", what, n ? "DIFFERENT" : "bit-identical", n);
  };
  auto run = [&](const double *t, int opt) {
    EdgeProfileLds kf{g, km, ec, cod.d + 3 * km, crx.d, xfx.d, cxa.d, xfa.d, (int)nCX, cry.d, yfx.d, cya.d, yfa.d, (int)nCY, t, opt};
#ifdef LAB_EDGE_WAVES
    int rc = launch_2w(Dim3{(unsigned)kf.nblocks(), 1, 1}, 2 * kFBuf, 0, kf);
#else
    int rc = launch(Dim3{(unsigned)kf.nblocks(), 1, 1}, 2 * kFBuf, 0, kf);
#endif
    if (rc) { std::fprintf(stderr, "launch failed %d\n", rc); std::exit(3); }
  };
  zero();
#ifdef FV3_LAB_EDGE_TABLE_ONLY
  run(tab.d, 0);
  const auto ref = grab();
#else
  run(nullptr, 0);
  const auto ref = grab();
  lab::time_it("edge_profile, rows built by the wavefront (round 5)", reps, none, [&] { run(nullptr, 0); }, bytes);
#endif
  for (int opt = 0; opt < 4; opt++) {
    zero();
    run(tab.d, opt);
    const auto o = grab();
    char label[128];
    std::snprintf(label, sizeof label, "edge_profile, rows from the host's table, opt %d", opt);
    lab::time_it(label, reps, none, [&] { run(tab.d, opt); }, bytes);
    same(label, ref, o);
  }
  // km not a multiple of 8 minus 1, and a small one: the bottom row sits elsewhere in its lane
  return 0;
}
