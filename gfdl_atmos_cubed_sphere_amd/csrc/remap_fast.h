// remap_fast.h -- Lagrangian_to_Eulerian (model/fv_mapz.F90:56-845) with the column in LDS and the LEVELS ACROSS THE LANES
// (the default where it is built; the slab kernels of remap_kernels.h take everything else: FV3_MI355X_REMAP_LDS).
//
// Why: RemapFields runs one thread per (column, field) with k sequential and keeps the spline's work arrays (a1, q, gam, a2..a4) in
// HBM slabs: 7 slab words written and read back per cell and field against 2 algorithmic ones, two or three dependent sweeps of 127
// levels per thread -- 7 % of the remap's own roofline (profiles/r03).  Here a workgroup owns 16 consecutive columns of a row:
//   * a field is read once with full 128-byte segments and laid out [column][level] in LDS, its coordinates beside it;
//   * the cubic-spline interface values (scalar_profile :572-623 / cs_profile :941-1016, fv_operators.F90) are ONE tridiagonal
//     system of km + 1 rows: tridiag_rows of nh_fast.h (8 rows per lane, 16 lanes per column, Moebius / affine scans);
//   * everything else is the parity kernel's own code on LDS operands, one thread per (column, level): the large-scale constraints
//     on the interface values, cs_cell (the subgrid limiters) and the search-and-integrate loop of map_scalar / map1_ppm /
//     mapn_tracer -- every target layer finds its first source layer by itself (the reference's k0 is the smallest l with
//     pe1(l+1) >= pe2(k), see map_target) and adds the source layers in the reference's order;
//   * T_v, delz and sphum stay in registers for what the remap ends with (delp, pk, peln, pkz, the conversion of pt, :426-503,
//     :793-841), so the remap of a column is one kernel; the D-grid winds are a second one (their own coordinates, :530-573).
// BIT-IDENTICAL to the parity kernels (round 4): the elimination runs in the reference's own order -- a lane runs its 8 rows from the
// value its neighbour hands it, round after round, until no hand-over changes any more (spline(): the recurrences forget, 3-6 rounds).
// Built for: kord_tm < 0, every kord in 8..10 or 13..15, remap_te off, km <= 127; dry and (nonhydrostatic) moist_kappa / use_cond
// thermodynamics, flagstruct%fill (fillz on the remapped tracers).  Anything else takes the slab kernels.
#pragma once

#include "nh_fast.h"
#include "remap_kernels.h"

namespace fv3 {


// a thread's kIt global loads first, all in flight together (clamped addresses, no branch, no LDS store in between: the compiler
// cannot move a load across a store to LDS it cannot prove disjoint), then the stores
// (tl: the thread's index, opaque to the optimizer once per loop -- fresh_tid -- so that the (column, level) addresses of a phase are
// formed in that phase: as common subexpressions of the whole kernel they stayed alive through every field and were spilled)
#ifdef FV3_HOST_EMU
#define FV3_LOAD_LOOP(it) for (int it = 0, tl = tid; it < kIt; it++)
#else
#define FV3_LOAD_LOOP(it) _Pragma("unroll") for (int it = 0, tl = fresh_tid(tid); it < kIt; it++)
#endif

// A column of an LDS array: row r at [r + 2], r = -2 .. 16 L + 1, then one byte per row (a4_form).  The pitch depends on the levels per
// lane L (RLay<L>): L = 8 -> 132 rows + 16 doubles of codes = 148 (= 20 mod 32: the 16 columns x 4 levels a wavefront stages at a time
// spread over the banks), 78 KB per workgroup, two per CU; L = 5 -> 84 rows + 10 = 94, 50 KB per workgroup, THREE per CU.  (A chunked
// layout -- 8 rows at 9 doubles, as nh_fast.h -- took a quarter of the bank conflicts of the spline away and was slower for its index
// arithmetic: profiles/r04_remap_layouts.txt; it is no longer carried.)
template <int L>
struct RLay {
  static constexpr int RL = 16 * L;                       // rows the lanes hold
  static constexpr int RC = 2 + RL + 2;                   // rows of a column
  static constexpr int RP = RC + (RL + 7) / 8;            // + the codes
  static constexpr int RQS = RC;                          // C1 carries no codes: the bottom value of w sits in their place
  static constexpr int RBuf = kFC * RP;
  // behind the four arrays: the level coefficients ak, bk (128 each), the surface pressures of the 16 columns and the kord of the first
  // 64 tracers -- as global loads inside the staging loops each of them cost a full `s_waitcnt vmcnt(0)`
  static constexpr int TabAk = 4 * RBuf, TabBk = TabAk + 128, TabPs = TabBk + 128, TabKord = TabPs + kFC;
  static constexpr int Lds = TabKord + 32;                // doubles per workgroup
};
FV3_HD int rix(int r) { return r + 2; }
FV3_HD int rixn(int r) { return r + 2; }
// row k (1-based) of a column
struct RCol {
  double *p;
  FV3_HD double &operator[](int k) const { return p[rix(k - 1)]; }
};
struct RColC {
  const double *p;
  FV3_HD double operator[](int k) const { return p[rix(k - 1)]; }
};
// One byte per row (chunked: in the ninth double of its chunk; linear: behind the rows): which expression the subgrid limiters formed
// the curvature a4 of cell k with (cs_cell / cs_limit, remap_kernels.h) -- map_target forms a4 again with that expression instead of keeping
// a third array of cell coefficients in LDS: 0: 3 (2 a1 - (a2 + a3)); 1: 6 a1 - 3 (a2 + a3); 2: 3 (a2 - a1); 3: 3 (a3 - a1)
FV3_HD unsigned char *a4_form_ptr(double *col, int k, int rc) { return reinterpret_cast<unsigned char *>(col + rc) + (k - 1); }
FV3_HD int a4_form(const double *col, int k, int rc) { return reinterpret_cast<const unsigned char *>(col + rc)[k - 1]; }
FV3_HD double a4_of(int form, double a1, double a2, double a3) {
  switch (form) {
    case 0: return 3. * (2. * a1 - (a2 + a3));
    case 1: return 6. * a1 - 3. * (a2 + a3);
    case 2: return 3. * (a2 - a1);
    default: return 3. * (a3 - a1);
  }
}
FV3_HD int a4_form_of(double a4, double a1, double a2, double a3) {   // the first expression that gives a4 bit for bit (one of them formed it)
  if (a4 == 3. * (2. * a1 - (a2 + a3))) return 0;
  if (a4 == 6. * a1 - 3. * (a2 + a3)) return 1;
  if (a4 == 3. * (a2 - a1)) return 2;
  return 3;
}
// Indices into the fields are 32-bit (a field of one tile is far below 2^32 bytes; the dispatch checks): with 64-bit indices the compiler
// keeps the eight (column, level) addresses of a thread for every array alive as 64-bit register pairs through the whole kernel and
// spills them (88 registers at two wavefronts per SIMD, 2.8 GB of scratch traffic per call); a 32-bit index is one register, shared by the
// arrays of a layout, and goes into the scalar-base form of the load (ix_t: nh_fast.h)
constexpr int kRKordMax = 64;   // kord of the first 64 tracers sits in LDS (ints)

// L: levels per lane (16 lanes a column).  8 holds km <= 127; 5 (80 rows) holds km <= 79 -- C96 / C768 L79 columns then leave no lane
// idle, where 8 left 38 % of the rows empty (the remap of BASELINE config 5's block: 133 -> 85 ms)
#ifdef FV3_HOST_EMU
template <int L>
inline vd vlin_ld(const double *buf, int col0, int q) {      // row (lane & 15) * L + q of the lane's column; any q with row >= -2
  vd x;
  FV3_LANE_LOOP x.v[l] = buf[((l >> 4) + col0) * RLay<L>::RP + rixn((l & 15) * L + q)];
  return x;
}
template <int L>
inline void vlin_st(double *buf, int col0, int q, const vd &x) {
  FV3_LANE_LOOP buf[((l >> 4) + col0) * RLay<L>::RP + rixn((l & 15) * L + q)] = x.v[l];
}
template <int L>
inline vb vrow_lt(int q, int k) { vb r; FV3_LANE_LOOP r.v[l] = (l & 15) * L + q < k; return r; }
template <int L>
inline vb vrow_eq(int q, int k) { vb r; FV3_LANE_LOOP r.v[l] = (l & 15) * L + q == k; return r; }
template <int L>
inline vd vcol_lds(const double *p, int col0) { vd x; FV3_LANE_LOOP x.v[l] = p[((l >> 4) + col0) * RLay<L>::RP]; return x; }
#else
template <int L>
__device__ __forceinline__ vd vlin_ld(const double *buf, int col0, int q) {
  const int l = (int)(threadIdx.x & 63);
  return buf[((l >> 4) + col0) * RLay<L>::RP + rixn((l & 15) * L + q)];
}
template <int L>
__device__ __forceinline__ void vlin_st(double *buf, int col0, int q, vd x) {
  const int l = (int)(threadIdx.x & 63);
  buf[((l >> 4) + col0) * RLay<L>::RP + rixn((l & 15) * L + q)] = x;
}
template <int L>
__device__ __forceinline__ vb vrow_lt(int q, int k) { return (int)(threadIdx.x & 15) * L + q < k; }
template <int L>
__device__ __forceinline__ vb vrow_eq(int q, int k) { return (int)(threadIdx.x & 15) * L + q == k; }
template <int L>
__device__ __forceinline__ vd vcol_lds(const double *p, int col0) { return p[((int)((threadIdx.x & 63) >> 4) + col0) * RLay<L>::RP]; }
#endif

// kords the streamed form of the mapping loop handles (cs_cell): scalar_profile / cs_profile without the two-cell limiters of 11, 12
FV3_HD bool kord_fast(int kord) {
  const int a = kord < 0 ? -kord : kord;
  return kord > 7 && a >= 8 && a <= 15 && a != 11 && a != 12;
}

// One target layer of the search-and-integrate loop (fv_operators.F90:93-132 == :188-227 == :402-441; tracer_form: :277-335) with
// the arithmetic of map_col.  The reference searches from k0, the source layer the previous target layer ended in, and takes the
// first l with pe1(l) <= pe2(k) <= pe1(l+1): that is the smallest l with pe1(l+1) >= pe2(k) (the previous layer ended at pe2(k), in
// the first layer whose lower edge is not above it), found here from the guess l = k.  pe1, a1: 1-based columns in LDS; a2, a3: the
// limited edge values of every source cell (cs_cell), a4 formed here again with the expression the limiters used (a4_form: the value
// the reference stores, bit for bit).  The quotients through one reciprocal and a Markstein correction: the values of `/` (remap_kernels.h div_rn).
FV3_HD double map_target(const RColC pe1, const RColC a1, const RColC a2, const RColC a3, int km, int rc, bool tracer_form, int k,
                         double p2t, double p2b) {
  constexpr double r3 = 1. / 3., r23 = 2. / 3.;
  int l = k;
  while (l > 1 && pe1[l] >= p2t) l--;
  while (l < km && pe1[l + 1] < p2t) l++;
  const double p1t = pe1[l], p1b = pe1[l + 1];
  const double dp1 = p1b - p1t, rdp1 = rcp_rn(dp1);
  const double pl = div_rn(p2t - p1t, dp1, rdp1);
  const double b2 = a2[l], b3 = a3[l], b4 = a4_of(a4_form(a1.p, l, rc), a1[l], b2, b3);
  if (p2b <= p1b) {
    const double pr = div_rn(p2b - p1t, dp1, rdp1);
    if (tracer_form) {
      double fac1 = pr + pl;
      const double fac2 = r3 * (pr * fac1 + pl * pl);
      fac1 = 0.5 * fac1;
      return b2 + (b4 + b3 - b2) * fac1 - b4 * fac2;
    }
    return b2 + 0.5 * (b4 + b3 - b2) * (pr + pl) - b4 * r3 * (pr * (pr + pl) + pl * pl);
  }
  double qsum;
  if (tracer_form) {
    const double dp = p1b - p2t;
    double fac1 = 1. + pl;
    const double fac2 = r3 * (1. + pl * fac1);
    fac1 = 0.5 * fac1;
    qsum = dp * (b2 + (b4 + b3 - b2) * fac1 - b4 * fac2);
  } else {
    qsum = (p1b - p2t) * (b2 + 0.5 * (b4 + b3 - b2) * (1. + pl) - b4 * (r3 * (1. + pl * (1. + pl))));
  }
  for (int m = l + 1; m <= km; m++) {
    const double mt = pe1[m], mb = pe1[m + 1];
    if (p2b > mb) {
      qsum = qsum + (mb - mt) * a1[m];
    } else {
      const double dp = p2b - mt, dm = mb - mt;
      const double esl = div_rn(dp, dm, rcp_rn(dm));
      const double m2 = a2[m], m3 = a3[m], m4 = a4_of(a4_form(a1.p, m, rc), a1[m], m2, m3);
      if (tracer_form) {
        const double fac1 = 0.5 * esl, fac2 = 1. - r23 * esl;
        qsum = qsum + dp * (m2 + fac1 * (m3 - m2 + m4 * fac2));
      } else {
        qsum = qsum + dp * (m2 + 0.5 * esl * (m3 - m2 + m4 * (1. - r23 * esl)));
      }
      break;
    }
  }
  const double dp2 = p2b - p2t;
  return div_rn(qsum, dp2, rcp_rn(dp2));
}

// the machinery both kernels share: a workgroup's 16 columns in the four LDS arrays
template <int L>
struct RemapFastCoreT {
  using Lay = RLay<L>;
  static constexpr int RL = Lay::RL, kRP = Lay::RP;   // rows of a column the lanes hold, pitch of a column
  int km;
  int probe = 0;   // timing probe (FV3_MI355X_REMAP_PROBE, tools/remap_time.py; WRONG results): 1 no spline, 2 no constraints, 4 no
                   // subgrid limiters, 8 no mapping loop
  // tools/lab (-DFV3_LAB_TRACE): clock64() of every wavefront at the phase boundaries of one block, in the order they are passed
  long long *trace = nullptr;
  mutable int tn = 0;
#ifdef FV3_LAB_TRACE
  FV3_D void mark(int tid) const { if (trace && (tid & 63) == 0) trace[(tid >> 6) * 128 + (tn < 127 ? tn : 127)] = clock64(); tn++; }
#else
  FV3_D void mark(int) const {}
#endif
  static constexpr int kIt = kFC * RL / kNT;   // (column, level) pairs per thread

  FV3_D static RCol col(double *buf, int c) { return RCol{buf + c * kRP}; }          // [k], k 1-based
  FV3_D static RColC colc(const double *buf, int c) { return RColC{buf + c * kRP}; }
  FV3_D static double &at(double *buf, int c, int r) { return buf[c * kRP + rix(r)]; }   // row r >= 0 of column c

  // pads of a coordinate array: rows -2, -1 and km+1 .. 129 continue with unit steps (layer thickness 1: the padded rows of
  // the system stay regular); of a field array: zeros.  Called by one thread per column after the real rows are in place.
  FV3_D void pad_coord(double *buf, int c, int nrow) const {
    double *p = buf + c * kRP;
    p[rixn(-1)] = p[rix(0)] - 1.; p[rixn(-2)] = p[rix(0)] - 2.;
    for (int r = nrow; r < RL + 2; r++) p[rix(r)] = p[rix(nrow - 1)] + (double)(r - nrow + 1);
  }
  FV3_D void pad_field(double *buf, int c, int nrow) const {
    double *p = buf + c * kRP;
    p[rixn(-1)] = 0.; p[rixn(-2)] = 0.;
    for (int r = nrow; r < RL + 2; r++) p[rix(r)] = 0.;
  }

  // interface values of the cubic spline: raw q(1 .. km+1) into Q.  iv = -2: scalar_profile / cs_profile with the bottom value qs
  // given (:572-595 / :941-964), otherwise :597-623 / :967-1016.  One wavefront = 4 columns; between barriers of the caller.
  //
  // The reference's elimination, in the reference's order, bit for bit.  Row k (interface k; lane = (k - 1) / 8) of both forms is
  //     bet_k = B_k - S_k gam_(k-1),   gam_k = N_k / bet_k,   q_k = (R_k - S_k q_(k-1)) / bet_k      (forward, k = 1 .. km + 1)
  //     q_k = q_k - G_k q_(k+1)                                                                     (backward)
  // with S = 1 in the interior (x * 1 is exact), S = a_bot in the bottom closure, and B = 1, S = 0 where the reference assigns a value
  // outright (the top closure, q(km+1) = qs, the padded rows): then bet = 1 and the quotients are N and R themselves.  Every
  // expression is the parity kernel's (remap_kernels.h profile_col); the quotients through a correctly rounded reciprocal and a
  // Markstein correction = the values of `/`.  A lane runs its 8 rows from the value the lane above hands it; all lanes do so at once,
  // ROUND AFTER ROUND, each round from the hand-overs of the one before.  After round r the lanes 0 .. r - 1 hold the sequential
  // values (lane 0 starts from the closure), so 16 rounds are the sequential sweep; and when a round leaves every hand-over as it
  // found it the state is the sequential one already (induction from lane 0) -- these recurrences forget (d gam_k / d gam_(k-1) =
  // N / bet^2 ~ 0.07, d q_k / d q_(k-1) = 1 / bet ~ 0.27), so that happens after 3 - 6 rounds.
  FV3_D void spline(const double *C1, const double *A1, double *Q, const double *QS, int iv, int wv) const {
    const int c0 = wv * 4;
    vd B[L], N[L], R[L], S[L], G[L], bet[L], rb[L], x[L];
    {
      vd e[L + 1], av[L], dpv[L];
      for (int q = 0; q <= L; q++) e[q] = vlin_ld<L>(C1, c0, q);
      const vd em1 = vlin_ld<L>(C1, c0, -1), em2 = vlin_ld<L>(C1, c0, -2);
      for (int q = 0; q < L; q++) {
        av[q] = vlin_ld<L>(A1, c0, q);
        dpv[q] = e[q + 1] - e[q];
      }
      const vd am1v = vlin_ld<L>(A1, c0, -1), am2v = vlin_ld<L>(A1, c0, -2);
      const vd dpm1v = e[0] - em1, dpm2v = em1 - em2;
      const vd qs = QS ? vcol_lds<L>(QS, c0) : vd(0.0);            // QS[column * kRP]
      vd grv[L];
      for (int q = 0; q < L; q++) grv[q] = vdivq(q > 0 ? dpv[q - 1] : dpm1v, dpv[q]);      // dp(k-1) / dp(k) of row k = r + 1
      const vd gr_up = row_shr<1>(grv[L - 1], 1.0);                                       // ... of the row above the lane's first
      (void)dpm2v;
      for (int q = 0; q < L; q++) {
        const vd a_m1 = q > 0 ? av[q - 1] : am1v, a_m2 = q > 1 ? av[q - 2] : (q == 1 ? am1v : am2v);
        const vb first = vrow_eq<L>(q, 0), pad = !vrow_lt<L>(q, km + 1), bot = vrow_eq<L>(q, km);
        const vd gr = grv[q];
        const vd bi = 2. + gr + gr;
        if (iv == -2) {
          const vb lastc = vrow_eq<L>(q, km - 1);
          const vb given = first || bot || pad;
          const vd rhs = 3. * (a_m1 + av[q]);
          B[q] = vsel(given, vd(1.0), bi);
          S[q] = vsel(given, vd(0.0), vd(1.0));
          N[q] = vsel(first, vd(0.5), vsel(bot || pad || lastc, vd(0.0), gr));
          R[q] = vsel(first, 1.5 * av[q], vsel(bot, qs, vsel(pad, vd(0.0), vsel(lastc, rhs - gr * qs, rhs))));
          G[q] = vsel(vrow_lt<L>(q, km - 1), vd(1.0), vd(0.0));      // back substitution on rows k <= km - 1, with gam(k+1) = this row's
        } else {
          // top row: grat = dp(2) / dp(1) (rows 0 and 1 are the lane's own); bottom row: d4 = dp(km-1) / dp(km)
          const vd g1 = vdivq(dpv[1], dpv[0]);
          const vd d4b = q > 0 ? grv[q - 1] : gr_up;   // dp(km-1) / dp(km) at the bottom row: the quotient of the row above
          const vd a_bot = 1. + d4b * (d4b + 1.5);
          B[q] = vsel(first, g1 * (g1 + 0.5), vsel(bot, d4b * (d4b + 0.5), vsel(pad, vd(1.0), bi)));
          S[q] = vsel(first || pad, vd(0.0), vsel(bot, a_bot, vd(1.0)));
          N[q] = vsel(first, 1. + g1 * (g1 + 1.5), vsel(bot || pad, vd(0.0), gr));
          R[q] = vsel(first, (g1 + g1) * (g1 + 1.) * av[0] + av[1],
                      vsel(bot, 2. * d4b * (d4b + 1.) * a_m1 + a_m2, vsel(pad, vd(0.0), 3. * (a_m1 + gr * av[q]))));
          G[q] = vsel(vrow_lt<L>(q, km), vd(1.0), vd(0.0));          // rows k <= km, with gam(k)
        }
      }
    }
    constexpr int kRounds = 16;
#ifdef FV3_ROUND_STATS
    extern long g_round_stats[3][20];
#define FV3_RS(c, r) g_round_stats[c][r]++
#else
#define FV3_RS(c, r)
#endif
    {                                     // bet, 1 / bet, gam: G[q] becomes this row's multiplier of the back substitution
      vd gin(0.27);
      vd gam[L];
      for (int rnd = 0; rnd < kRounds; rnd++) {
        vd g = gin;
        for (int q = 0; q < L; q++) {
          bet[q] = B[q] - S[q] * g;
          rb[q] = vrecip(bet[q]);
          g = vdiv_r(N[q], bet[q], rb[q]);
          gam[q] = g;
        }
        const vd gnew = row_shr<1>(g, 0.27);
        const bool moved = vany_ne(gnew, gin);
        gin = gnew;
        if (!moved) { FV3_RS(0, rnd); break; }
      }
      for (int q = 0; q < L; q++) G[q] = G[q] * gam[q];
    }
    {                                     // forward substitution
      vd qin(0.0);
      for (int rnd = 0; rnd < kRounds; rnd++) {
        vd y = qin;
        for (int q = 0; q < L; q++) {
          y = vdiv_r(R[q] - S[q] * y, bet[q], rb[q]);
          x[q] = y;
        }
        const vd qnew = row_shr<1>(y, 0.0);
        const bool moved = vany_ne(qnew, qin);
        qin = qnew;
        if (!moved) { FV3_RS(1, rnd); break; }
      }
    }
    {                                     // back substitution
      vd xin(0.0);
      vd y[L];
      for (int q = 0; q < L; q++) y[q] = x[q];
      for (int rnd = 0; rnd < kRounds; rnd++) {
        vd xn = xin;
        for (int q = L - 1; q >= 0; q--) {
          xn = y[q] - G[q] * xn;
          x[q] = xn;
        }
        const vd xnew = row_shl<1>(xn, 0.0);
        const bool moved = vany_ne(xnew, xin);
        xin = xnew;
        if (!moved) { FV3_RS(2, rnd); break; }
      }
    }
    for (int q = 0; q < L; q++) vlin_st<L>(Q, c0, q, x[q]);
  }

  // large-scale constraints on the interface values (:643-680 / :1037-1073), in place in Q; one thread per (column, interface)
  FV3_D void constrain(const double *A1, double *Q, int iv, int ak, int tid) const {
    for (int idx = tid; idx < kFC * RL; idx += kNT) {
      const int col = idx / RL, k = idx % RL + 1;
      if (k < 2 || k > km) continue;
      const RColC a1 = colc(A1, col);   // a1[k], 1-based
      const RCol q = RemapFastCoreT::col(Q, col);
      const double w_m1 = a1[k - 1], w_0 = a1[k];
      double qc = q[k];
      if (k == 2 || k == km) {
        const double v = dmin(qc, dmax(w_m1, w_0));
        qc = dmax(v, dmin(w_m1, w_0));
      } else {
        const double gm = w_m1 - a1[k - 2], gp = a1[k + 1] - w_0;
        if (ak >= 14 || gm * gp > 0.) {
          qc = dmin(qc, dmax(w_m1, w_0));
          qc = dmax(qc, dmin(w_m1, w_0));
        } else if (gm > 0.) {
          qc = dmax(qc, dmin(w_m1, w_0));
        } else {
          qc = dmin(qc, dmax(w_m1, w_0));
          if (iv == 0) qc = dmax(0., qc);
        }
      }
      q[k] = qc;
    }
  }

  // subgrid limiters (cs_cell) of every source cell, then the mapping loop of every target layer; one thread per (column, k) for
  // both.  The limited edge values a2 / a3 replace Q / C2 for the duration of the mapping loop (the thread keeps the two target
  // pressures of its layer in registers and puts C2 back afterwards); on return Q holds the remapped layer means.
  FV3_D void map_all(double *C1, double *C2, double *A1, double *Q, bool is_scalar, int iv, int ak, double qmin, bool tracer_form,
                     int tid) const {
    double r2[kIt], r3v[kIt], pt2[kIt], pb2[kIt];
    int form[kIt];
    const ProfCfg pc{km, iv, ak, is_scalar, qmin, true};
    for (int it = 0; it < kIt; it++) {
      const int idx = tid + it * kNT, col = idx / RL, k = idx % RL + 1;
      r2[it] = r3v[it] = pt2[it] = pb2[it] = 0.;
      form[it] = 0;
      if (k > km) continue;
      const RColC a1 = colc(A1, col), q = colc(Q, col), t2 = colc(C2, col);
      double a2v = q[k], a3v = q[k + 1], a4v = 0.;
      if (!(probe & 4))
        cs_cell(pc, k, a2v, a3v, k - 2 >= 1 ? a1[k - 2] : 0., k - 1 >= 1 ? a1[k - 1] : 0., a1[k], k + 1 <= km ? a1[k + 1] : 0.,
                k + 2 <= km ? a1[k + 2] : 0., a4v);
      r2[it] = a2v; r3v[it] = a3v;
      form[it] = a4_form_of(a4v, a1[k], a2v, a3v);
      pt2[it] = t2[k]; pb2[it] = t2[k + 1];
    }
    mark(tid);   // limiters done
    FV3_SYNC_LDS();
    for (int it = 0; it < kIt; it++) {
      const int idx = tid + it * kNT, col = idx / RL, k = idx % RL + 1;
      if (k > km) continue;
      at(Q, col, k - 1) = r2[it];
      at(C2, col, k - 1) = r3v[it];
      *a4_form_ptr(A1 + col * kRP, k, Lay::RC) = (unsigned char)form[it];
    }
    FV3_SYNC_LDS();
    for (int it = 0; it < kIt; it++) {
      const int idx = tid + it * kNT, col = idx / RL, k = idx % RL + 1;
      if (k > km) continue;
      if (!(probe & 8)) r2[it] = map_target(colc(C1, col), colc(A1, col), colc(Q, col), colc(C2, col), km, Lay::RC, tracer_form, k, pt2[it], pb2[it]);
    }
    FV3_SYNC_LDS();
    for (int it = 0; it < kIt; it++) {
      const int idx = tid + it * kNT, col = idx / RL, k = idx % RL + 1;
      if (k > km) continue;
      at(Q, col, k - 1) = r2[it];
      at(C2, col, k - 1) = pt2[it];
      if (k == km) at(C2, col, km) = pb2[it];
    }
  }

  // a whole field: A1 and (for iv = -2) QS are staged, C1 / C2 hold the coordinates; on return (after its last barrier) Q holds the
  // remapped layer means.  `pre` issues the global loads of whatever the caller stages next: they are in flight while the limiters and
  // the mapping loop run (the barriers order LDS only)
  template <class Pre>
  FV3_D void remap_field(double *C1, double *C2, double *A1, double *Q, const double *QS, bool is_scalar, int iv, int kord,
                         double qmin, bool tracer_form, int tid, Pre &&pre) const {
    const int ak = kord < 0 ? -kord : kord;
    mark(tid);   // field start
    if (!(probe & 1)) { FV3_WAVE_FOR(wv) { spline(C1, A1, Q, QS, iv, wv); } }
    mark(tid);   // spline done
    FV3_SYNC_LDS();
    if (!(probe & 2)) constrain(A1, Q, iv, ak, tid);
    FV3_SYNC_LDS();
    mark(tid);   // constrained
    pre();
    map_all(C1, C2, A1, Q, is_scalar, iv, ak, qmin, tracer_form, tid);
    FV3_SYNC_LDS();
    mark(tid);   // mapped
  }
};

// ---- the scalars of a column: T_v, w, delz, the tracers, omega; delp, pk, peln, pkz, ps and the conversion of pt ------------------
// Staging (round 4, second half): the fields go through the four LDS arrays one after the other, and with __syncthreads() between the
// phases every field paid its store latency and then the next field's load latency (the kernel with every computation switched off
// ran at 1.3 TB/s, profiles/r04_v1_remap_probe.txt).  Now (a) the barriers order LDS only (FV3_SYNC_LDS: every global location this
// kernel reads after writing it is read by the thread that wrote it), so stores drain behind the next phase, and (b) the inputs of the
// NEXT field are loaded into registers just before the mapping loop of the current one (`pre` of remap_field: 8 or 16 doubles per
// thread, not live during the spline, which is where the register budget is tight) and go to LDS when that loop is done.
// MOIST: thermostruct%moist_kappa / use_cond (nonhydrostatic; fv3_set_moist): the temperature transform with cappa from moist_cv of
// the un-remapped tracers (:212-219), pkz with cappa from the remapped ones (:463-478), q_con / cappa written on the way, and the
// conversion of the last step with the condensates (:806-811) -- the slab kernels' expressions (remap_kernels.h) per (column, level)
template <bool HYDRO, bool MOIST = false, int L = kFL>   // the flags at compile time: the other branches' loads and registers are not carried
struct RemapFastScalars {
  static_assert(!(HYDRO && MOIST), "moist_kappa / use_cond are nonhydrostatic branches");
  Grid g;
  int km;
  RemapPar p;
  const double *ak, *bk;
  const int *kord_tr;   // device, nq
  const double *pe, *ws;
  double *ps, *delp, *pkz, *pk, *delz, *pt, *peln, *w, *q, *omga;
  int probe = 0;
  int opt = 1;     // A / B switch of tools/lab/remap_lab.hip (on in the library): 1 XCD-contiguous column blocks
  long long *trace = nullptr;   // tools/lab (-DFV3_LAB_TRACE): RemapFastCoreT::mark of block trace_blk
  int trace_blk = -1;

  FV3_HD int nblocks_x() const { return (g.nx + kFC - 1) / kFC; }

  FV3_D void operator()(int bx, int by, int, int tid, double *lds) const {
    using Core = RemapFastCoreT<L>;
    using Lay = RLay<L>;
    constexpr int kIt = Core::kIt, RL = Core::RL, kRP = Lay::RP, kRBuf = Lay::RBuf;
    double *C1 = lds, *C2 = lds + kRBuf, *A1 = lds + 2 * kRBuf, *Q = lds + 3 * kRBuf, *QS = C1 + Lay::RQS;   // QS[column * kRP]
    if (opt & 1) xcd_block(bx, by, nblocks_x(), g.ny);   // neighbouring column blocks share 128-byte lines: one L2 for both (nh_fast.h)
    const Core core{km, probe, by * nblocks_x() + bx == trace_blk ? trace : nullptr};
    core.mark(tid);   // block start
    const int i0 = g.is + bx * kFC, j = g.js + by;
    const int ncol = (g.nx - bx * kFC < kFC) ? g.nx - bx * kFC : kFC;
    const ix_t nA = g.nA(), nCC = g.nCC();
    const ix_t o0 = (ix_t)g.iA(i0, j), occ0 = (ix_t)g.iCC(i0, j);
    const ix_t peb0 = (ix_t)(j - (g.js - 1)) * (g.nx + 2) * (km + 1) + (i0 - (g.is - 1));
    const ix_t lnb0 = (ix_t)(j - g.js) * g.nx * (km + 1) + (i0 - g.is);
    const double k1k = p.rdgas / p.cv_air, rrg = -p.rdgas / p.grav, akap = p.akap;
    const int akt = p.kord_tm < 0 ? -p.kord_tm : p.kord_tm;
    auto clampc = [&](int col) { return col < ncol ? col : ncol - 1; };
    // staging map: idx -> (col = idx & 15, k0 = idx >> 4): 16 consecutive threads read 16 consecutive columns of a level
    // (what the end of the kernel needs of the remapped fields -- T_v, delz, sphum -- is written to pt / delz / q as it is formed and
    // read back by the same thread there; the log of the new interface pressures is formed again: kept in registers through the
    // remap of every field these 32 doubles per thread were spilled to scratch, 404 B per lane)
    double *AK = lds + Lay::TabAk, *BK = lds + Lay::TabBk, *PS = lds + Lay::TabPs;
    int *KT = reinterpret_cast<int *>(lds + Lay::TabKord);
    double nx0[kIt], nx1[kIt];   // the inputs of the next field, in flight during the mapping loop of the current one
    // the pressure coordinate's source interfaces (pe) and the first field on it (w; hydrostatic: the first tracer)
    // (every prefetch is unconditional -- a load under a branch is waited for at the join; where there is nothing to fetch the
    // address is that of a field that exists)
    auto load_pressures = [&]() __attribute__((always_inline)) {
      const double *f1 = HYDRO ? (p.nq > 0 ? q : pt) : w;
      FV3_LOAD_LOOP(it) {
        const int idx = tl + it * kNT, col = idx & (kFC - 1), k0 = idx >> 4, cc = clampc(col);
        const int ki = k0 <= km ? k0 : km, kc = k0 < km ? k0 : km - 1;
        nx0[it] = pe[peb0 + (ix_t)ki * (g.nx + 2) + cc];
        nx1[it] = f1[(ix_t)kc * nA + o0 + cc];
      }
    };
    auto load_tracer = [&](int iq) __attribute__((always_inline)) {
      const double *qq = p.nq > 0 ? q + (size_t)(iq < p.nq ? iq : p.nq - 1) * nA * km : pt;
      FV3_LOAD_LOOP(it) {
        const int idx = tl + it * kNT, col = idx & (kFC - 1), k0 = idx >> 4;
        nx0[it] = qq[(ix_t)(k0 < km ? k0 : km - 1) * nA + o0 + clampc(col)];
      }
    };
    // ---- log-pressure coordinates of T_v (:340-345, :363-368): C1 = peln, C2 = pn2; the layer means: the temperature transform
    //      (:200-229) level by level ----
    double v_cap[MOIST ? kIt : 1];   // MOIST: cap / (1 - cap) of the thread's cells for the temperature transform
    if constexpr (MOIST) {
      for (int it = 0, tl = fresh_tid(tid); it < kIt; it++) {
        const int idx = tl + it * kNT, col = idx & (kFC - 1), k0 = idx >> 4, cc = clampc(col);
        const ix_t o3 = (ix_t)(k0 < km ? k0 : km - 1) * nA + o0 + cc;
        v_cap[it] = 0.;
        if (p.moist_kappa) {   // :212-219, from the tracers as they are before the remap
          double qc;
          const double cvm = moist_cv(p, q + o3, (size_t)nA * km, qc);
          const double cap = p.rdgas / (p.rdgas + cvm / (1. + p.r_vir * q[(size_t)(p.sphum - 1) * nA * km + o3]));
          if (k0 < km && col < ncol) {
            p.q_con[o3] = qc;
            p.cappa[o3] = cap;
          }
          v_cap[it] = cap / (1. - cap);
        }
      }
    }
    {
      double v_pl[kIt], v_t[kIt], v_a[kIt], v_b[kIt], v_c[HYDRO ? kIt : 1];
#ifndef FV3_HOST_EMU
      // the tables: loaded in front of the fields (the counter of outstanding loads is in order), stored to LDS behind them
      const int tn = tid & 127, tkk = tn <= km ? tn : km;
      const double t_ak = ak[tkk], t_bk = bk[tkk], t_ps = pe[peb0 + (ix_t)km * (g.nx + 2) + clampc(tid & (kFC - 1))];
      const int t_kord = p.nq > 0 ? kord_tr[(tid & (kRKordMax - 1)) < p.nq ? (tid & (kRKordMax - 1)) : p.nq - 1] : 0;
#endif
      FV3_LOAD_LOOP(it) {
        const int idx = tl + it * kNT, col = idx & (kFC - 1), k0 = idx >> 4, cc = clampc(col);
        const int ki = k0 <= km ? k0 : km, kc = k0 < km ? k0 : km - 1;     // interface / cell row, clamped
        const ix_t o3 = (ix_t)kc * nA + o0 + cc, c3 = (ix_t)kc * nCC + occ0 + cc;
        v_pl[it] = peln[lnb0 + (ix_t)ki * g.nx + cc];
        v_t[it] = pt[o3];
        if (HYDRO) {
          v_a[it] = pk[c3 + nCC]; v_b[it] = pk[c3]; v_c[it] = peln[lnb0 + (ix_t)(kc + 1) * g.nx + cc];
        } else {
          v_a[it] = delp[o3]; v_b[it] = delz[c3];
        }
      }
#ifdef FV3_HOST_EMU
      for (int n = 0; n < 128; n++) {
        const int kk = n <= km ? n : km;
        AK[n] = ak[kk]; BK[n] = bk[kk];
        if (n < kFC) PS[n] = pe[peb0 + (ix_t)km * (g.nx + 2) + clampc(n)];   // the surface pressure of column n
        if (n < kRKordMax) KT[n] = p.nq > 0 ? kord_tr[n < p.nq ? n : p.nq - 1] : 0;
      }
#else
      if (tid < 128) { AK[tn] = t_ak; BK[tn] = t_bk; }
      if (tid < kFC) PS[tid] = t_ps;
      if (tid < kRKordMax) KT[tid] = t_kord;
#endif
      FV3_SYNC_LDS();   // AK, BK, PS, KT
      for (int it = 0, tl = fresh_tid(tid); it < kIt; it++) {
        const int idx = tl + it * kNT, col = idx & (kFC - 1), k0 = idx >> 4, cc = clampc(col);
        if (k0 <= km) {
          Core::at(C1, col, k0) = v_pl[it];
          const double v_ps1 = PS[col];
          Core::at(C2, col, k0) = (k0 == 0 || k0 == km) ? v_pl[it] : dlog(AK[k0] + BK[k0] * v_ps1);
          if (k0 == 0) ps[o0 + cc] = v_ps1;   // :298-300
        }
        if (k0 < km) {
          double t = v_t[it];
          if (HYDRO)
            t = t * (v_a[it] - v_b[it]) / (akap * (v_c[HYDRO ? it : 0] - v_pl[it]));
          else if (MOIST && p.moist_kappa)
            t = t * dexp(v_cap[MOIST ? it : 0] * dlog(rrg * v_a[it] / v_b[it] * t));
          else
            t = t * dexp(k1k * dlog(rrg * v_a[it] / v_b[it] * t));
          Core::at(A1, col, k0) = t;
        }
      }
    }
    FV3_SYNC_LDS();
    for (int col = tid; col < kFC; col += kNT) {
      core.pad_coord(C1, col, km + 1);
      core.pad_coord(C2, col, km + 1);
      core.pad_field(A1, col, km);
      QS[col * kRP] = HYDRO ? 0. : ws[occ0 + clampc(col)];
    }
    FV3_SYNC_LDS();
    core.remap_field(C1, C2, A1, Q, nullptr, true, 1, akt, p.t_min, false, tid, [&]() __attribute__((always_inline)) { load_pressures(); });
    for (int it = 0, tl = fresh_tid(tid); it < kIt; it++) {
      const int idx = tl + it * kNT, col = idx & (kFC - 1), k0 = idx >> 4;
      if (k0 < km && col < ncol) pt[(ix_t)k0 * nA + o0 + col] = Core::at(Q, col, k0);   // T_v for now
    }
    // ---- omega on the last step (:432-443, :506-526): interpolated in the old log-p coordinate (C1) to the centres of the new
    //      layers (C2); pe3(k) = omga(k-1), pe3(1) = 0 in A1 ----
    if (p.last_step) {
      FV3_SYNC_LDS();
      {
        double v_o[kIt];
        FV3_LOAD_LOOP(it) {
          const int idx = tl + it * kNT, col = idx & (kFC - 1), k0 = idx >> 4;
          const int kc = k0 < 1 ? 0 : (k0 <= km ? k0 - 1 : km - 1);
          v_o[it] = omga[(ix_t)kc * nA + o0 + clampc(col)];
        }
        for (int it = 0, tl = fresh_tid(tid); it < kIt; it++) {
          const int idx = tl + it * kNT, col = idx & (kFC - 1), k0 = idx >> 4;
          if (k0 <= km) Core::at(A1, col, k0) = k0 == 0 ? 0. : v_o[it];
        }
      }
      FV3_SYNC_LDS();
      double om[kIt];
      for (int it = 0, tl = fresh_tid(tid); it < kIt; it++) {
        const int idx = tl + it * kNT, col = idx / RL, n = idx % RL + 1;
        om[it] = 0.;
        if (n > km) continue;
        const RColC e = Core::colc(C1, col), t2 = Core::colc(C2, col), p3 = Core::colc(A1, col);
        const double mid = 0.5 * (t2[n] + t2[n + 1]);
        int k = n;                                  // the reference's first k (from k_next) with e(k) <= mid <= e(k+1)
        while (k > 1 && e[k] >= mid) k--;
        while (k < km && e[k + 1] < mid) k++;
        om[it] = p3[k] + (p3[k + 1] - p3[k]) * (mid - e[k]) / (e[k + 1] - e[k]);
      }
      FV3_SYNC_LDS();
      for (int it = 0, tl = fresh_tid(tid); it < kIt; it++) {
        const int idx = tl + it * kNT, col = idx / RL, n = idx % RL + 1;
        if (n <= km) Core::at(Q, col, n - 1) = om[it];
      }
      FV3_SYNC_LDS();
      for (int it = 0, tl = fresh_tid(tid); it < kIt; it++) {
        const int idx = tl + it * kNT, col = idx & (kFC - 1), k0 = idx >> 4;
        if (k0 < km && col < ncol) omga[(ix_t)k0 * nA + o0 + col] = Core::at(Q, col, k0);
      }
    }
    FV3_SYNC_LDS();
    // ---- pressure coordinates for everything else: C1 = pe, C2 = pe2 (:318-322); nx0 = pe, nx1 = the first field on them ----
    for (int it = 0, tl = fresh_tid(tid); it < kIt; it++) {
      const int idx = tl + it * kNT, col = idx & (kFC - 1), k0 = idx >> 4;
      if (k0 <= km) {
        Core::at(C1, col, k0) = nx0[it];
        const double v_ps1 = PS[col];
        Core::at(C2, col, k0) = (k0 == 0) ? p.ptop : (k0 == km ? v_ps1 : AK[k0] + BK[k0] * v_ps1);
      }
      if ((!HYDRO || p.nq > 0) && k0 < km) Core::at(A1, col, k0) = nx1[it];
    }
    FV3_SYNC_LDS();
    for (int col = tid; col < kFC; col += kNT) {
      core.pad_coord(C1, col, km + 1);
      core.pad_coord(C2, col, km + 1);
      core.pad_field(A1, col, km);
    }
    if (!HYDRO) {
      FV3_SYNC_LDS();
      // ---- w (:400-411): iv = -2, the bottom value ws; delz and delp come in behind its mapping loop ----
      core.remap_field(C1, C2, A1, Q, QS, false, -2, p.kord_wz, 0., false, tid, [&]() __attribute__((always_inline)) {
        FV3_LOAD_LOOP(it) {
          const int idx = tl + it * kNT, col = idx & (kFC - 1), k0 = idx >> 4, cc = clampc(col);
          const int kc = k0 < km ? k0 : km - 1;
          nx0[it] = delz[(ix_t)kc * nCC + occ0 + cc];
          nx1[it] = delp[(ix_t)kc * nA + o0 + cc];
        }
      });
      for (int it = 0, tl = fresh_tid(tid); it < kIt; it++) {
        const int idx = tl + it * kNT, col = idx & (kFC - 1), k0 = idx >> 4;
        if (k0 < km && col < ncol) w[(ix_t)k0 * nA + o0 + col] = Core::at(Q, col, k0);
        // ---- delz (:292, :412-423): the specific volume -delz / delp in, delz = -q2 dp2 out ----
        if (k0 < km) Core::at(A1, col, k0) = -nx0[it] / nx1[it];
      }
      FV3_SYNC_LDS();
      core.remap_field(C1, C2, A1, Q, nullptr, false, 1, akt, 0., false, tid, [&]() __attribute__((always_inline)) { load_tracer(0); });
      for (int it = 0, tl = fresh_tid(tid); it < kIt; it++) {
        const int idx = tl + it * kNT, col = idx & (kFC - 1), k0 = idx >> 4;
        if (k0 < km) {
          const double dzn = -Core::at(Q, col, k0) * (Core::at(C2, col, k0 + 1) - Core::at(C2, col, k0));
          if (col < ncol) delz[(ix_t)k0 * nCC + occ0 + col] = dzn;
          if (p.nq > 0) Core::at(A1, col, k0) = nx0[it];
        }
      }
    }
    // ---- the tracers (:380-397): A1 holds tracer iq when its turn comes, tracer iq + 1 comes in behind its mapping loop ----
    for (int iq = 0; iq < p.nq; iq++) {
      double *qq = q + (size_t)iq * nA * km;
      const bool more = iq + 1 < p.nq;
      FV3_SYNC_LDS();
      core.remap_field(C1, C2, A1, Q, nullptr, true, 0, iq < kRKordMax ? KT[iq] : kord_tr[iq], 0., p.nq > 5, tid, [&]() __attribute__((always_inline)) { load_tracer(iq + 1); });
      if (p.fill) {   // flagstruct%fill: fillz (fv_fill.F90:34-137) on the remapped column, sequential in k as in the reference -- one
                      // thread per column, and only for a column that holds a negative value at all (dp2(k) = pe2(k+1) - pe2(k) from C2)
                for (int col = tid; col < kFC; col += kNT) {
          double *qc = Q + col * kRP + rix(0);
          const double *e2 = C2 + col * kRP + rix(0);
          bool neg = false;
          for (int k = 0; k < km; k++) neg = neg || qc[k] < 0.;
          if (neg) fillz_col(km, qc, 1, [&](int k) { return e2[k] - e2[k - 1]; });
        }
        FV3_SYNC_LDS();
      }
      for (int it = 0, tl = fresh_tid(tid); it < kIt; it++) {
        const int idx = tl + it * kNT, col = idx & (kFC - 1), k0 = idx >> 4;
        if (k0 < km) {
          const double v = Core::at(Q, col, k0);
          if (col < ncol) qq[(ix_t)k0 * nA + o0 + col] = v;
          if (more) Core::at(A1, col, k0) = nx0[it];
        }
      }
    }
    FV3_SYNC_LDS();
    core.mark(tid);   // fields done
    // ---- the new interfaces: pn2 -> A1, pk2 -> Q (:340-345); then delp, pk, peln, pkz and pt of every layer (:426-503, :793-841);
    //      the loads of the last phase (T_v, delz, sphum as this thread stored them) are issued in front of the logarithms ----
    {
      const bool need_qv = p.last_step && p.last_step != 2 && !p.adiabatic && p.sphum > 0;
      const double *qs_ = need_qv ? q + (size_t)(p.sphum - 1) * nA * km : pt;
      double v_pl[kIt], v_pk[kIt], v_t[kIt], v_dz[kIt], v_q[kIt];
      FV3_LOAD_LOOP(it) {
        const int idx = tl + it * kNT, col = idx & (kFC - 1), k0 = idx >> 4, cc = clampc(col);
        const int ke = (k0 == 0) ? 0 : km, kc = k0 < km ? k0 : km - 1;
        v_pl[it] = peln[lnb0 + (ix_t)ke * g.nx + cc];
        v_pk[it] = pk[(ix_t)ke * nCC + occ0 + cc];
        v_t[it] = pt[(ix_t)kc * nA + o0 + cc];
        v_dz[it] = HYDRO ? 1. : delz[(ix_t)kc * nCC + occ0 + cc];
        v_q[it] = qs_[(ix_t)kc * nA + o0 + cc];
      }
      for (int it = 0, tl = fresh_tid(tid); it < kIt; it++) {
        const int idx = tl + it * kNT, col = idx & (kFC - 1), k0 = idx >> 4;
        if (k0 <= km) {
          double pn, pkv;
          if (k0 == 0 || k0 == km) {
            pn = v_pl[it];
            pkv = v_pk[it];
          } else {
            pn = dlog(AK[k0] + BK[k0] * PS[col]);
            pkv = dexp(akap * pn);
            if (col < ncol) {
              peln[lnb0 + (ix_t)k0 * g.nx + col] = pn;
              pk[(ix_t)k0 * nCC + occ0 + col] = pkv;
            }
          }
          Core::at(A1, col, k0) = pn;
          Core::at(Q, col, k0) = pkv;
        }
      }
      FV3_SYNC_LDS();
      for (int it = 0, tl = fresh_tid(tid); it < kIt; it++) {
        const int idx = tl + it * kNT, col = idx & (kFC - 1), k0 = idx >> 4;
        if (k0 >= km || col >= ncol) continue;
        const RColC t2 = Core::colc(C2, col), pn = Core::colc(A1, col), pk2 = Core::colc(Q, col);   // [k], 1-based: row k0 is [k0 + 1]
        const ix_t o3 = (ix_t)k0 * nA + o0 + col, c3 = (ix_t)k0 * nCC + occ0 + col;
        const double dp2 = t2[k0 + 2] - t2[k0 + 1];
        delp[o3] = dp2;
        const double tv = v_t[it];
        double pkzv;
        if (HYDRO) {
          pkzv = (pk2[k0 + 2] - pk2[k0 + 1]) / (akap * (pn[k0 + 2] - pn[k0 + 1]));
        } else if (MOIST && p.moist_kappa) {   // :463-478: cappa from the REMAPPED tracers (this thread stored them)
          double qc;
          const double cvm = moist_cv(p, q + o3, (size_t)nA * km, qc);
          const double cap = p.rdgas / (p.rdgas + cvm / (1. + p.r_vir * q[(size_t)(p.sphum - 1) * nA * km + o3]));
          p.q_con[o3] = qc;
          p.cappa[o3] = cap;
          pkzv = dexp(cap * dlog(rrg * dp2 / v_dz[it] * tv));
        } else {
          pkzv = dexp(akap * dlog(rrg * dp2 / v_dz[it] * tv));
        }
        pkz[c3] = pkzv;
        double tn = tv;
        if (p.last_step == 2) {              // the energy fixer follows: T_v stays, fv3_remap_finish converts (:793-821)
        } else if (p.last_step) {            // :793-821 (dtmp = 0)
          if (MOIST && p.use_cond) {         // :806-811
            double qc;
            const double cvm = moist_cv(p, q + o3, (size_t)nA * km, qc);
            tn = (tn + 0. / cvm * pkzv) / ((1. + p.r_vir * q[(size_t)(p.sphum - 1) * nA * km + o3]) * (1. - qc));
          } else if (!p.adiabatic) {
            tn = (tn + 0. / (HYDRO ? p.cp : p.cv_air) * pkzv) / (1. + p.r_vir * (need_qv ? v_q[it] : 0.));
          }
        } else {
          tn = tn / pkzv;                    // :833-841
        }
        pt[o3] = tn;
      }
    }
    core.mark(tid);   // block end
  }
};

// ---- the D-grid winds (:530-573): u on (is:ie, js:je+1), v on (is:ie+1, js:je), each on the mean pressure of its two cells ----------
template <int WHICH, int L = kFL>   // 0: u, 1: v
struct RemapFastWind {
  Grid g;
  int km;
  int kord_mt;
  const double *ak, *bk, *pe;
  double *f;
  int probe = 0;
  int opt = 1;

  FV3_HD int ncols_row() const { return WHICH == 0 ? g.nx : g.nx + 1; }
  FV3_HD int nrows() const { return WHICH == 0 ? g.ny + 1 : g.ny; }
  FV3_HD int nblocks_x() const { return (ncols_row() + kFC - 1) / kFC; }

  FV3_D void operator()(int bx, int by, int, int tid, double *lds) const {
    using Core = RemapFastCoreT<L>;
    using Lay = RLay<L>;
    constexpr int kIt = Core::kIt, kRBuf = Lay::RBuf;
    const Core core{km, probe};
    double *C1 = lds, *C2 = lds + kRBuf, *A1 = lds + 2 * kRBuf, *Q = lds + 3 * kRBuf;
    double *AK = lds + Lay::TabAk, *BK = lds + Lay::TabBk;
    if (opt & 1) xcd_block(bx, by, nblocks_x(), nrows());
    const int i0 = g.is + bx * kFC, j = g.js + by;
    const int ncol = (ncols_row() - bx * kFC < kFC) ? ncols_row() - bx * kFC : kFC;
    const ix_t fs = WHICH == 0 ? g.nU() : g.nV();
    const ix_t f0 = WHICH == 0 ? (ix_t)g.iU(i0, j) : (ix_t)g.iV(i0, j);
    auto PE = [&](int ii, int k0, int jj) {
      return pe[(ix_t)(jj - (g.js - 1)) * (g.nx + 2) * (km + 1) + (ix_t)k0 * (g.nx + 2) + (ii - (g.is - 1))];
    };
    {
      double v_a[kIt], v_b[kIt], v_sa[kIt], v_sb[kIt], v_f[kIt];
#ifndef FV3_HOST_EMU
      const int tn = tid & 127, tkk = tn <= km ? tn : km;
      const double t_ak = ak[tkk], t_bk = bk[tkk];
#endif
      FV3_LOAD_LOOP(it) {
        const int idx = tid + it * kNT, col = idx & (kFC - 1), k0 = idx >> 4, cc = col < ncol ? col : ncol - 1;
        const int i = i0 + cc, i2 = WHICH == 0 ? i : i - 1, j2 = WHICH == 0 ? j - 1 : j;
        const int ki = k0 <= km ? k0 : km, kc = k0 < km ? k0 : km - 1;
        v_a[it] = PE(i2, ki, j2); v_b[it] = PE(i, ki, j);
        v_sa[it] = PE(i2, km, j2); v_sb[it] = PE(i, km, j);
        v_f[it] = f[(ix_t)kc * fs + f0 + cc];
      }
#ifdef FV3_HOST_EMU
      for (int n = 0; n < 128; n++) {
        const int kk = n <= km ? n : km;
        AK[n] = ak[kk]; BK[n] = bk[kk];
      }
#else
      if (tid < 128) { AK[tn] = t_ak; BK[tn] = t_bk; }
#endif
      FV3_SYNC_LDS();   // AK, BK
      for (int it = 0; it < kIt; it++) {
        const int idx = tid + it * kNT, col = idx & (kFC - 1), k0 = idx >> 4;
        if (k0 <= km) {
          const double psum = v_sa[it] + v_sb[it];
          Core::at(C1, col, k0) = (k0 == 0) ? v_b[it] : 0.5 * (v_a[it] + v_b[it]);
          const double bkh = 0.5 * BK[k0];
          Core::at(C2, col, k0) = (WHICH == 1 && k0 == 0) ? AK[0] : AK[k0] + bkh * psum;
        }
        if (k0 < km) Core::at(A1, col, k0) = v_f[it];
      }
    }
    FV3_SYNC_LDS();
    for (int col = tid; col < kFC; col += kNT) {
      core.pad_coord(C1, col, km + 1);
      core.pad_coord(C2, col, km + 1);
      core.pad_field(A1, col, km);
    }
    FV3_SYNC_LDS();
    core.remap_field(C1, C2, A1, Q, nullptr, false, -1, kord_mt, 0., false, tid, []() {});
    for (int it = 0; it < kIt; it++) {
      const int idx = tid + it * kNT, col = idx & (kFC - 1), k0 = idx >> 4;
      if (k0 < km && col < ncol) f[(ix_t)k0 * fs + f0 + col] = Core::at(Q, col, k0);
    }
  }
};

// three workgroups per CU at 5 levels per lane (50 KB of LDS each): the register budget of three wavefronts per SIMD
template <bool HYDRO, bool MOIST>
struct tile_waves<RemapFastScalars<HYDRO, MOIST, 5>> { static constexpr int value = 3; };
template <int WHICH>
struct tile_waves<RemapFastWind<WHICH, 5>> { static constexpr int value = 3; };

}  // namespace fv3
