// remap_lab.hip -- Lagrangian_to_Eulerian's levels-across-the-lanes kernels (csrc/remap_fast.h) on a synthetic C384 L127 tile with 4 tracers,
// outside the library: variants timed against each other and compared bit for bit.  tools/lab/build.sh remap_lab; run on the GPU box.
//   remap_lab [nx] [km] [reps] [nq]
#include "lab_common.h"

#include "../../gfdl_atmos_cubed_sphere_amd/csrc/fv3_launch.h"
#include "../../gfdl_atmos_cubed_sphere_amd/csrc/remap_fast.h"

using namespace fv3;
using lab::DevArr;

struct State {
  Grid g{};
  int km, nq;
  RemapPar rp{};
  DevArr ak, bk, pe, ws, ps, delp, pkz, pk, delz, pt, peln, w, q, omga, u, v;
  DevArr delp0, pt0, delz0, w0, q0, omga0, peln0, pk0, u0, v0;
  int *kord_tr = nullptr;
};

static void make_state(State &s, int nx, int km, int nq) {
  Grid &g = s.g;
  std::memset(&g, 0, sizeof(g));
  g.is = 1; g.ie = nx; g.js = 1; g.je = nx;
  g.isd = 1 - NG; g.ied = nx + NG; g.jsd = 1 - NG; g.jed = nx + NG;
  g.npx = nx + 1; g.npy = nx + 1; g.npz = km;
  g.nid = nx + 2 * NG; g.njd = nx + 2 * NG; g.nx = nx; g.ny = nx;
  g.grid_type = 4;
  s.km = km; s.nq = nq;
  const double GRAV = 9.80665, RDGAS = 287.05, CP = 1004.6, KAPPA = RDGAS / CP, PTOP = 300.;
  RemapPar &p = s.rp;
  p.last_step = 0; p.hydrostatic = 0; p.adiabatic = 1; p.nq = nq; p.kord_mt = 9; p.kord_wz = 9; p.kord_tm = -9; p.sphum = nq ? 1 : 0;
  p.akap = KAPPA; p.ptop = PTOP; p.rdgas = RDGAS; p.grav = GRAV; p.cv_air = CP - RDGAS; p.r_vir = 0.6077; p.cp = CP; p.t_min = 184.;
  p.fill = 0;
  const size_t nA = g.nA(), nCC = g.nCC(), nU = g.nU(), nV = g.nV();
  s.ak.alloc(km + 1); s.bk.alloc(km + 1);
  s.pe.alloc((size_t)(nx + 2) * (nx + 2) * (km + 1)); s.ws.alloc(nCC); s.ps.alloc(nA); s.delp.alloc(nA * km); s.pkz.alloc(nCC * km);
  s.pk.alloc(nCC * (km + 1)); s.delz.alloc(nCC * km); s.pt.alloc(nA * km); s.peln.alloc(nCC * (km + 1)); s.w.alloc(nA * km);
  s.q.alloc(nA * km * (nq ? nq : 1)); s.omga.alloc(nA * km); s.u.alloc(nU * km); s.v.alloc(nV * km);
  lab::Rng r(31);
  std::vector<double> sig(km + 1);
  for (int k = 0; k <= km; k++) {
    sig[k] = std::pow((double)k / km, 1.5);
    s.ak.h[k] = PTOP * (1. - sig[k]);
    s.bk.h[k] = sig[k];
  }
  // columns on the (nx + 2)^2 box of pe; the A fields take the same values where they overlap
  std::vector<double> pecol((size_t)(nx + 2) * (nx + 2) * (km + 1));
  for (int jj = 0; jj < nx + 2; jj++)
    for (int ii = 0; ii < nx + 2; ii++) {
      const double ps = 1.0e5 * (1. + 0.01 * r.sym());
      double pe = PTOP;
      for (int k = 0; k <= km; k++) {
        s.pe.h[(size_t)jj * (nx + 2) * (km + 1) + (size_t)k * (nx + 2) + ii] = pe;
        if (k < km) pe += (ps - PTOP) * (sig[k + 1] - sig[k]) * (1. + 0.04 * r.sym());
      }
    }
  for (int j = g.jsd; j <= g.jed; j++)
    for (int i = g.isd; i <= g.ied; i++) {
      const size_t c = g.iA(i, j);
      const int ii = std::min(std::max(i, 0), nx + 1), jj = std::min(std::max(j, 0), nx + 1);
      double zs = 50. * r.uni();
      for (int k = 0; k < km; k++) {
        const double pt_ = s.pe.h[(size_t)jj * (nx + 2) * (km + 1) + (size_t)k * (nx + 2) + ii];
        const double pb_ = s.pe.h[(size_t)jj * (nx + 2) * (km + 1) + (size_t)(k + 1) * (nx + 2) + ii];
        const double dp = pb_ - pt_, pm = dp / std::log(pb_ / pt_);
        const double T = 300. - 60. * (1. - sig[k + 1]) + 2. * r.sym();
        const double pt = T * std::pow(pm, -KAPPA);
        s.delp.h[(size_t)k * nA + c] = dp;
        s.pt.h[(size_t)k * nA + c] = pt;
        s.w.h[(size_t)k * nA + c] = 1.5 * r.sym();
        s.omga.h[(size_t)k * nA + c] = r.sym();
        for (int n = 0; n < nq; n++) { const double x = r.uni(); s.q.h[((size_t)n * km + k) * nA + c] = x * x * x * (n == 0 ? 0.02 : 1.); }
        if (i >= 1 && i <= nx && j >= 1 && j <= nx) {
          const size_t cc = g.iCC(i, j);
          s.delz.h[(size_t)k * nCC + cc] = -dp / GRAV * RDGAS * pt * std::pow(pm, KAPPA - 1.) * (1. + 0.02 * r.sym());
        }
      }
      (void)zs;
      if (i >= 1 && i <= nx && j >= 1 && j <= nx) {
        const size_t cc = g.iCC(i, j);
        s.ws.h[cc] = 0.05 * r.sym();
        for (int k = 0; k <= km; k++) {
          const double pe = s.pe.h[(size_t)jj * (nx + 2) * (km + 1) + (size_t)k * (nx + 2) + ii];
          s.peln.h[(size_t)(j - 1) * nx * (km + 1) + (size_t)k * nx + (i - 1)] = std::log(pe);
          s.pk.h[(size_t)k * nCC + cc] = std::exp(KAPPA * std::log(pe));
        }
      }
    }
  for (size_t n = 0; n < s.u.n; n++) s.u.h[n] = 10. + 5. * r.sym();
  for (size_t n = 0; n < s.v.n; n++) s.v.h[n] = -3. + 5. * r.sym();
  for (DevArr *a : {&s.ak, &s.bk, &s.pe, &s.ws, &s.delp, &s.pk, &s.delz, &s.pt, &s.peln, &s.w, &s.q, &s.omga, &s.u, &s.v}) a->up();
  auto keep = [](DevArr &dst, DevArr &src) { dst.alloc(src.n); HC(hipMemcpy(dst.d, src.d, src.n * 8, hipMemcpyDeviceToDevice)); };
  keep(s.delp0, s.delp); keep(s.pt0, s.pt); keep(s.delz0, s.delz); keep(s.w0, s.w); keep(s.q0, s.q); keep(s.omga0, s.omga);
  keep(s.peln0, s.peln); keep(s.pk0, s.pk); keep(s.u0, s.u); keep(s.v0, s.v);
  std::vector<int> kt(64, 9);
  HC(hipMalloc(&s.kord_tr, 64 * sizeof(int)));
  HC(hipMemcpy(s.kord_tr, kt.data(), 64 * sizeof(int), hipMemcpyHostToDevice));
}

static void reset(State &s) {
  auto cp = [](DevArr &dst, DevArr &src) { HC(hipMemcpyAsync(dst.d, src.d, src.n * 8, hipMemcpyDeviceToDevice, 0)); };
  cp(s.delp, s.delp0); cp(s.pt, s.pt0); cp(s.delz, s.delz0); cp(s.w, s.w0); cp(s.q, s.q0); cp(s.omga, s.omga0);
  cp(s.peln, s.peln0); cp(s.pk, s.pk0); cp(s.u, s.u0); cp(s.v, s.v0);
}

struct Out { std::vector<double> ps, delp, pkz, pk, delz, pt, peln, w, q, omga, u, v; };
static Out grab(State &s) {
  HC(hipDeviceSynchronize());
  return Out{s.ps.get(), s.delp.get(), s.pkz.get(), s.pk.get(), s.delz.get(), s.pt.get(), s.peln.get(), s.w.get(), s.q.get(), s.omga.get(),
             s.u.get(), s.v.get()};
}
static bool same(const char *what, const Out &a, const Out &b) {
  size_t n = 0;
  n += lab::count_diff(a.ps, b.ps); n += lab::count_diff(a.delp, b.delp); n += lab::count_diff(a.pkz, b.pkz); n += lab::count_diff(a.pk, b.pk);
  n += lab::count_diff(a.delz, b.delz); n += lab::count_diff(a.pt, b.pt); n += lab::count_diff(a.peln, b.peln); n += lab::count_diff(a.w, b.w);
  n += lab::count_diff(a.q, b.q); n += lab::count_diff(a.omga, b.omga); n += lab::count_diff(a.u, b.u); n += lab::count_diff(a.v, b.v);
  std::printf("  %-44s %s (%zu words differ)\n", what, n ? "DIFFERENT" : "bit-identical", n);
  return n == 0;
}

template <int L>
static RemapFastScalars<false, false, L> make_sc(State &s, int last_step) {
  RemapPar rp = s.rp;
  rp.last_step = last_step;
  return RemapFastScalars<false, false, L>{s.g, s.km, rp, s.ak.d, s.bk.d, s.kord_tr, s.pe.d, s.ws.d, s.ps.d, s.delp.d, s.pkz.d, s.pk.d, s.delz.d,
                                           s.pt.d, s.peln.d, s.w.d, s.q.d, s.omga.d, 0};
}
template <class K>
static void go(const K &k, unsigned gx, unsigned gy, size_t lds_doubles) {
  int rc = launch_2w(Dim3{gx, gy, 1}, lds_doubles, 0, k);
  if (rc) { std::fprintf(stderr, "launch failed %d\n", rc); std::exit(3); }
}

int main(int argc, char **argv) {
  const int nx = argc > 1 ? std::atoi(argv[1]) : 384, km = argc > 2 ? std::atoi(argv[2]) : 127;
  const int reps = argc > 3 ? std::atoi(argv[3]) : 5, nq = argc > 4 ? std::atoi(argv[4]) : 4;
  State s;
  make_state(s, nx, km, nq);
  const double cells = (double)nx * nx * km;
  std::printf("remap_lab %d x %d x %d, %d tracers\n", nx, nx, km, nq);
  auto rs = [&] { reset(s); };
  constexpr int L = 8;
  const size_t lds = RLay<L>::Lds;
  const unsigned gx = (unsigned)((nx + kFC - 1) / kFC);
  auto sc = make_sc<L>(s, 0);
  RemapFastWind<0, L> wu{s.g, km, s.rp.kord_mt, s.ak.d, s.bk.d, s.pe.d, s.u.d};
  RemapFastWind<1, L> wv{s.g, km, s.rp.kord_mt, s.ak.d, s.bk.d, s.pe.d, s.v.d};
  auto all = [&](auto &k, auto &a, auto &b) {
    go(k, gx, (unsigned)nx, lds);
    go(a, (unsigned)a.nblocks_x(), (unsigned)a.nrows(), lds);
    go(b, (unsigned)b.nblocks_x(), (unsigned)b.nrows(), lds);
  };
  Out ref;
  {
    auto k0 = sc; auto a0 = wu; auto b0 = wv;
#ifdef LAB_OPT
    k0.opt = 0; a0.opt = 0; b0.opt = 0;
#endif
    reset(s); all(k0, a0, b0); ref = grab(s);
    double bytes = cells * 8. * (2 * (3 + nq) + 2 + 4 + 5);   // in and out of T_v, w, delz, tracers; u, v; pe, peln, pk in; delp, pkz, pk, peln, ps out
    lab::time_it("remap scalars (round 5)", reps, rs, [&] { go(k0, gx, (unsigned)nx, lds); });
    lab::time_it("remap winds u + v (round 5)", reps, rs, [&] { go(a0, (unsigned)a0.nblocks_x(), (unsigned)a0.nrows(), lds); go(b0, (unsigned)b0.nblocks_x(), (unsigned)b0.nrows(), lds); });
    lab::time_it("remap all three (round 5)", reps, rs, [&] { all(k0, a0, b0); }, bytes);
    for (int pr : {1, 2, 4, 8, 15}) {
      auto kp = k0; kp.probe = pr;
      char lb[96];
      std::snprintf(lb, sizeof lb, "remap scalars (round 5) probe %d (wrong results)", pr);
      lab::time_it(lb, reps, rs, [&] { go(kp, gx, (unsigned)nx, lds); });
    }
  }
#ifdef LAB_OPT
  for (int opt : {LAB_OPT}) {
    auto k1 = sc; auto a1 = wu; auto b1 = wv;
    k1.opt = opt; a1.opt = opt; b1.opt = opt;
    char lb[96];
    std::snprintf(lb, sizeof lb, "remap scalars opt %d", opt);
    lab::time_it(lb, reps, rs, [&] { go(k1, gx, (unsigned)nx, lds); });
    std::snprintf(lb, sizeof lb, "remap winds u + v opt %d", opt);
    lab::time_it(lb, reps, rs, [&] { go(a1, (unsigned)a1.nblocks_x(), (unsigned)a1.nrows(), lds); go(b1, (unsigned)b1.nblocks_x(), (unsigned)b1.nrows(), lds); });
    std::snprintf(lb, sizeof lb, "remap all three opt %d", opt);
    lab::time_it(lb, reps, rs, [&] { all(k1, a1, b1); });
    reset(s); all(k1, a1, b1); same(lb, ref, grab(s));
    {
      auto kl0 = make_sc<L>(s, 1); kl0.opt = 0;
      auto kl1 = make_sc<L>(s, 1); kl1.opt = opt;
      reset(s); go(kl0, gx, (unsigned)nx, lds); Out r0 = grab(s);
      reset(s); go(kl1, gx, (unsigned)nx, lds);
      std::snprintf(lb, sizeof lb, "remap scalars opt %d, last step", opt);
      same(lb, r0, grab(s));
    }
  }
#endif
#ifdef FV3_LAB_TRACE
  {
    long long *tr;
    HC(hipMalloc(&tr, 4 * 128 * 8));
    for (int blk : {4000}) {
      HC(hipMemset(tr, 0, 4 * 128 * 8));
      reset(s);
      auto k1 = sc; k1.trace = tr; k1.trace_blk = blk;
      go(k1, gx, (unsigned)nx, lds);
      HC(hipDeviceSynchronize());
      long long h[512];
      HC(hipMemcpy(h, tr, sizeof h, hipMemcpyDeviceToHost));
      std::printf("trace scalars block %d: cycles between marks (block start | per field: start, spline, constrained, limiters, mapped | fields done | end)\n", blk);
      for (int w = 0; w < 4; w++) {
        std::printf("  wave %d:", w);
        for (int n = 1; n < 128 && h[w * 128 + n]; n++) std::printf(" %lld", h[w * 128 + n] - h[w * 128 + n - 1]);
        std::printf("\n");
      }
    }
  }
#endif
  return 0;
}
