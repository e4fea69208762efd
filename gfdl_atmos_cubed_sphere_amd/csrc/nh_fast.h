// nh_fast.h -- the semi-implicit column solvers and edge_profile with the LEVELS ACROSS THE LANES, in the reference's order.
//
//   RiemFast<false>   Riem_Solver3   model/nh_core.F90:47-241   + SIM1_solver model/nh_utils.F90:1277-1394 / SIM_solver :1396-1537
//   RiemFast<true>    Riem_Solver_c  model/nh_utils.F90:323-480 + SIM1_solver
//   EdgeProfileLds    edge_profile   model/nh_utils.F90:1590-1696
//
// Why: the slab kernels (nh_kernels.h sim_column) run one thread per column with k sequential.  A 384 x 384 tile has 2 304 such
// wavefronts for 1 024 SIMDs: 2.25 dependent instruction streams per SIMD, six sweeps through HBM scratch slabs (31 word accesses per
// cell against 9 algorithmic) -- 10-13 % of their own roofline.  Here a 16-lane row of a wavefront owns ONE column, every
// lane 8 consecutive levels of it (km <= 127), a wavefront 4 columns, a workgroup 16 consecutive columns of a row:
//   * fields are read once with full 128-byte segments (16 columns x 8 B per level) and transposed through LDS; no scratch slabs;
//   * everything that is pointwise in k (three log, three exp per cell, the matrix coefficients) is evaluated with the slab kernel's
//     own expressions, 8 independent levels per lane;
//   * the recurrences in k run in the reference's own order: hand-over rounds between the lanes for those that forget (tridiag_rounds),
//     km / 8 + 1 rounds for the sums, one wavefront over LDS for the stiff w system (RiemFast::w_column) -- BIT-IDENTICAL to the slab
//     kernels (DESIGN 3c "Round 4").
// (Rounds 2 - 4 also carried a "tolerance mode" here -- the recurrences as blocked parallel scans, Stone's recursive doubling: same
// equations, different association, 1e-14 per call but 2e-12 in w after a whole dt_atmos, outside north_star's 1e-12.  Removed in
// round 5: the library has one mode, the one the oracle is held to bit for bit.)
#pragma once

#include "nh_kernels.h"
#include "spmd.h"

namespace fv3 {

// indices into the fields: 32 bits (a field of one tile is far below 2^32 bytes; the dispatch checks): one register instead of a pair per
// (column, level) address a thread keeps, shared by the arrays of a layout, and the scalar-base form of the load / store
using ix_t = unsigned;
constexpr int kFL = 8;     // levels per lane
constexpr int kFC = 16;    // columns per workgroup (4 per wavefront)
constexpr int kFS = kFL + 1;         // doubles per 8-level chunk in LDS: 9, so that the 16 lanes of a column hit 16 different bank pairs
constexpr int kFP = 16 * kFS + 1;   // doubles per column in an LDS transposition buffer (level k at (k / 8) * 9 + k % 8)
constexpr int kFBuf = kFC * kFP;
constexpr int kFNBuf = 4;           // transposition buffers per workgroup (all four inputs staged at once: one barrier)
FV3_HD int lds_lev(int k) { return (k >> 3) * kFS + (k & 7); }

#ifdef FV3_HOST_EMU
#define FV3_WAVE_FOR(wv) for (int wv = 0; wv < 4; wv++)
constexpr int kWvState = 4;
#define FV3_WVI(wv) (wv)
#define FV3_LANE_LOOP for (int l = 0; l < kW; l++)
template <int N>
inline vd row_shr(const vd &a, double fill) {   // lane l <- lane l - N of the same 16-lane row, `fill` where there is none
  vd r;
  FV3_LANE_LOOP r.v[l] = ((l & 15) >= N) ? a.v[l - N] : fill;
  return r;
}
template <int N>
inline vd row_shl(const vd &a, double fill) {
  vd r;
  FV3_LANE_LOOP r.v[l] = ((l & 15) + N <= 15) ? a.v[l + N] : fill;
  return r;
}
inline vd vlog(const vd &a) { vd r; FV3_LANE_LOOP r.v[l] = dlog(a.v[l]); return r; }
inline vd vexp(const vd &a) { vd r; FV3_LANE_LOOP r.v[l] = dexp(a.v[l]); return r; }
inline vd vrcp(const vd &a) { vd r; FV3_LANE_LOOP r.v[l] = 1. / a.v[l]; return r; }
inline vd vfma(const vd &a, const vd &b, const vd &c) { vd r; FV3_LANE_LOOP r.v[l] = __builtin_fma(a.v[l], b.v[l], c.v[l]); return r; }
inline vd vlds_ld(const double *buf, int col0, int q) {
  vd r;
  FV3_LANE_LOOP r.v[l] = buf[((l >> 4) + col0) * kFP + (l & 15) * kFS + q];
  return r;
}
inline void vlds_st(double *buf, int col0, int q, const vd &x) {
  FV3_LANE_LOOP buf[((l >> 4) + col0) * kFP + (l & 15) * kFS + q] = x.v[l];
}
inline void vlds_st_next_if(double *buf, int col0, int q, const vd &x, const vb &m) {   // level (lane's level q) + 1, where m
  FV3_LANE_LOOP if (m.v[l]) buf[((l >> 4) + col0) * kFP + lds_lev((l & 15) * kFL + q + 1)] = x.v[l];
}
inline vd vrow_ld(const double *t, int q, int n) {   // t[row of the lane], the row clamped to the table
  vd r;
  FV3_LANE_LOOP { const int k = (l & 15) * kFL + q; r.v[l] = t[k < n ? k : n - 1]; }
  return r;
}
inline vb vlevel_lt(int q, int k) { vb r; FV3_LANE_LOOP r.v[l] = (l & 15) * kFL + q < k; return r; }
inline vb vlevel_eq(int q, int k) { vb r; FV3_LANE_LOOP r.v[l] = (l & 15) * kFL + q == k; return r; }
inline vd vcol_ld(const double *p, long o0, int col0, int ncol) {   // p[o0 + column of the lane] (clamped to the block's last column)
  vd r;
  FV3_LANE_LOOP { const int c = (l >> 4) + col0; r.v[l] = p[o0 + (c < ncol ? c : ncol - 1)]; }
  return r;
}
#else
#define FV3_WAVE_FOR(wv) for (int wv = (int)(threadIdx.x >> 6), once_ = 1; once_; once_ = 0)
constexpr int kWvState = 1;
#define FV3_WVI(wv) 0
template <int N>
__device__ __forceinline__ vd row_shr(vd a, double fill) {   // DPP row_shr:N, lanes without a source keep `fill`
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(fill), __double2loint(a), 0x110 + N, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), __double2hiint(a), 0x110 + N, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int N>
__device__ __forceinline__ vd row_shl(vd a, double fill) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(fill), __double2loint(a), 0x100 + N, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), __double2hiint(a), 0x100 + N, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// fv3_log / fv3_exp (include/fv3_math.h) WITHOUT their special-case branches (zero / subnormal / negative / inf / NaN arguments, overflow
// and underflow of exp): the arguments here are pressures, densities and their logarithms times kappa -- finite, positive, normal.  The
// arithmetic of the main path is the same operation for operation, so the values are fv3_log's / fv3_exp's bit for bit; without the
// branches the 8 independent levels of a lane interleave instead of running one call after the other.
__device__ __forceinline__ vd vlog(vd x) {
  const double LN2_HI = 0x1.62e42feep-1, LN2_LO = 0x1.a39ef35793c76p-33;
  const long long ix = fv3m_bits(x);
  const long long tmp = ix - 0x3fe6a09e667f3bcdLL;
  const int k = (int)(tmp >> 52);
  const double m = fv3m_from_bits(ix - (long long)((unsigned long long)tmp & 0xfff0000000000000ULL));
  const double f = m - 1.0;
  const double den = 2.0 + f;                       // in [1.7, 3.5]: the Markstein quotient below IS the IEEE quotient
  const double s = vdiv_r(f, den, vrecip(den));
  const double z = s * s, w = z * z;
  double t1 = 0x1.8618618618618p-4;
  t1 = __builtin_fma(t1, w, 0x1.e1e1e1e1e1e1ep-4);
  t1 = __builtin_fma(t1, w, 0x1.3b13b13b13b14p-3);
  t1 = __builtin_fma(t1, w, 0x1.c71c71c71c71cp-3);
  t1 = __builtin_fma(t1, w, 0x1.999999999999ap-2);
  double t2 = 0x1.642c8590b2164p-4;
  t2 = __builtin_fma(t2, w, 0x1.af286bca1af28p-4);
  t2 = __builtin_fma(t2, w, 0x1.1111111111111p-3);
  t2 = __builtin_fma(t2, w, 0x1.745d1745d1746p-3);
  t2 = __builtin_fma(t2, w, 0x1.2492492492492p-2);
  t2 = __builtin_fma(t2, w, 0x1.5555555555555p-1);
  const double R = __builtin_fma(t1, w, t2 * z);
  const double hfsq = 0.5 * f * f;
  const double kd = (double)k;
  return kd * LN2_HI - ((hfsq - __builtin_fma(s, hfsq + R, kd * LN2_LO)) - f);
}
__device__ __forceinline__ vd vexp(vd x) {
  const double INV_LN2 = 0x1.71547652b82fep+0;
  const double LN2_HI = 0x1.62e42fefa39efp-1, LN2_LO = 0x1.abc9e3b39803fp-56;
  const double SHIFT = 0x1.8p52;
  const double t = x * INV_LN2 + SHIFT;
  const double kd = t - SHIFT;
  const int k = (int)kd;
  double r = __builtin_fma(-kd, LN2_HI, x);
  r = __builtin_fma(-kd, LN2_LO, r);
  double q = 0x1.6124613a86d09p-33;
  q = __builtin_fma(q, r, 0x1.1eed8eff8d898p-29);
  q = __builtin_fma(q, r, 0x1.ae64567f544e4p-26);
  q = __builtin_fma(q, r, 0x1.27e4fb7789f5cp-22);
  q = __builtin_fma(q, r, 0x1.71de3a556c734p-19);
  q = __builtin_fma(q, r, 0x1.a01a01a01a01ap-16);
  q = __builtin_fma(q, r, 0x1.a01a01a01a01ap-13);
  q = __builtin_fma(q, r, 0x1.6c16c16c16c17p-10);
  q = __builtin_fma(q, r, 0x1.1111111111111p-7);
  q = __builtin_fma(q, r, 0x1.5555555555555p-5);
  q = __builtin_fma(q, r, 0x1.5555555555555p-3);
  q = __builtin_fma(q, r, 0.5);
  const double y = 1.0 + __builtin_fma(r * r, q, r);
  const int k1 = k >> 1, k2 = k - k1;
  return y * fv3m_from_bits((long long)(1023 + k1) << 52) * fv3m_from_bits((long long)(1023 + k2) << 52);
}
// 1/x to < 1 ulp (v_rcp_f64 + two Newton steps): the solves are tolerance-mode arithmetic
__device__ __forceinline__ vd vrcp(vd b) {
  double y = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-b, y, 1.0);
  return __builtin_fma(y, e, y);
}
__device__ __forceinline__ vd vfma(vd a, vd b, vd c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ vd vlds_ld(const double *buf, int col0, int q) {
  const int l = (int)(threadIdx.x & 63);
  return buf[((l >> 4) + col0) * kFP + (l & 15) * kFS + q];
}
__device__ __forceinline__ void vlds_st(double *buf, int col0, int q, vd x) {
  const int l = (int)(threadIdx.x & 63);
  buf[((l >> 4) + col0) * kFP + (l & 15) * kFS + q] = x;
}
__device__ __forceinline__ void vlds_st_next_if(double *buf, int col0, int q, vd x, vb m) {
  const int l = (int)(threadIdx.x & 63);
  if (m) buf[((l >> 4) + col0) * kFP + lds_lev((l & 15) * kFL + q + 1)] = x;
}
__device__ __forceinline__ vd vrow_ld(const double *t, int q, int n) {
  const int k = (int)(threadIdx.x & 15) * kFL + q;
  return t[k < n ? k : n - 1];
}
__device__ __forceinline__ vb vlevel_lt(int q, int k) { return (int)(threadIdx.x & 15) * kFL + q < k; }
__device__ __forceinline__ vb vlevel_eq(int q, int k) { return (int)(threadIdx.x & 15) * kFL + q == k; }
__device__ __forceinline__ vd vcol_ld(const double *p, long o0, int col0, int ncol) {
  const int c = (int)((threadIdx.x & 63) >> 4) + col0;
  return p[o0 + (c < ncol ? c : ncol - 1)];
}
#endif
#ifdef FV3_HOST_EMU
inline bool vany_ne(const vd &a, const vd &b) {               // some lane's bits differ
  FV3_LANE_LOOP if (fv3m_bits(a.v[l]) != fv3m_bits(b.v[l])) return true;
  return false;
}
inline vd row_shl1_v(const vd &a, const vd &fill) {           // lane l <- lane l + 1 of its row; the row's last lane keeps its own `fill`
  vd r;
  FV3_LANE_LOOP r.v[l] = ((l & 15) + 1 <= 15) ? a.v[l + 1] : fill.v[l];
  return r;
}
#else
__device__ __forceinline__ bool vany_ne(vd a, vd b) { return __builtin_amdgcn_ballot_w64(fv3m_bits(a) != fv3m_bits(b)) != 0; }
__device__ __forceinline__ vd row_shl1_v(vd a, vd fill) { return row_shl<1>(a, fill); }
#endif

// The column block a workgroup takes.  Consecutive workgroups of a launch go to the 8 XCDs in turn.  An A field has 3 halo columns, so
// no 16-column block starts on a 128-byte line and neighbouring blocks share one; as neighbours in the launch they ran on different
// XCDs and every shared line was fetched into two L2s (FETCH_SIZE 2.3 x the fields of Riem_Solver3, 2 x of the remap).  With this
// permutation each XCD owns one contiguous range of blocks (the last nb mod 8 blocks keep their place): counter reads of a call
// 1.39 -> 1.03 GB for Riem_Solver3, 1.23 -> 0.75 GB for Riem_Solver_c (tools/lab/riem_lab.hip, opt & 4).
FV3_HD void xcd_block(int &bx, int &by, int nbx, int nby) {
  const int flat = by * nbx + bx, chunk = (nbx * nby) >> 3;
  if (flat < chunk * 8) {
    const int lg = (flat & 7) * chunk + (flat >> 3);
    by = lg / nbx;
    bx = lg - by * nbx;
  }
}

// a / b, correctly rounded for operands in the normal range (spmd.h vdiv_r: one reciprocal + a Markstein correction, 8 instructions
// instead of the 14 of the compiler's IEEE division with its scale / fixup rescue): the same value as `a / b` here
FV3_D vd vdivq(const vd &a, const vd &b) { return vdiv_r(a, b, vrecip(b)); }

// Tridiagonal systems of one column, 8 rows per lane (rows beyond the system: a = c = d = 0, b = 1; a of the first row and c of the last
// row are 0):   a_k x_{k-1} + b_k x_k + c_k x_{k+1} = d_k,   in the REFERENCE'S ORDER, bit for bit (RiemFast<CG, true>; the idea of remap_fast.h spline()): row k is
//     bet_k = b_k - a_k gam_k,   gam_(k+1) = c_k / bet_k,   y_k = (d_k - a_k y_(k-1)) / bet_k        (downwards)
//     x_k = y_k - gam_(k+1) x_(k+1)                                                                  (upwards)
// with a_k = 1 (x * 1 is exact) or 0 (first row, padded rows: b = 1, c = d = 0).  A lane runs its 8 rows from the value the lane above
// hands it, all lanes at once, round after round; after round r lanes 0 .. r-1 hold the sequential values, and a round that leaves every
// hand-over as it found it has reached them everywhere.  For a system whose elimination forgets (diagonally dominant: the pp system of
// SIM1_solver, d gam_k / d gam_(k-1) = c / bet^2 ~ 0.1) that is after 3 - 4 rounds.  Quotients: a correctly rounded reciprocal and a
// Markstein correction = the values of `/` (spmd.h vrecip / vdiv_r = remap_kernels.h rcp_rn / div_rn).
FV3_D void tridiag_rounds(const vd *a, const vd *b, const vd *c, const vd *d, vd *x, const int kRounds = 16) {
  vd bet[kFL], rb[kFL], gam[kFL], y[kFL];
  {
    vd gin(0.25);
    for (int rnd = 0; rnd < kRounds; rnd++) {
      vd g = gin;
      for (int q = 0; q < kFL; q++) {
        bet[q] = b[q] - a[q] * g;
        rb[q] = vrecip(bet[q]);
        g = vdiv_r(c[q], bet[q], rb[q]);
        gam[q] = g;
      }
      const vd gnew = row_shr<1>(g, 0.25);
      const bool moved = vany_ne(gnew, gin);
      gin = gnew;
      if (!moved) break;
    }
  }
  {
    vd yin(0.0);
    for (int rnd = 0; rnd < kRounds; rnd++) {
      vd v = yin;
      for (int q = 0; q < kFL; q++) {
        v = vdiv_r(d[q] - a[q] * v, bet[q], rb[q]);
        y[q] = v;
      }
      const vd ynew = row_shr<1>(v, 0.0);
      const bool moved = vany_ne(ynew, yin);
      yin = ynew;
      if (!moved) break;
    }
  }
  {
    vd xin(0.0);
    for (int rnd = 0; rnd < kRounds; rnd++) {
      vd xn = xin;
      for (int q = kFL - 1; q >= 0; q--) {
        xn = y[q] - gam[q] * xn;
        x[q] = xn;
      }
      const vd xnew = row_shl<1>(xn, 0.0);
      const bool moved = vany_ne(xnew, xin);
      xin = xnew;
      if (!moved) break;
    }
  }
}

// edge_profile with the LEVELS ACROSS THE LANES and the elimination in the reference's order: BIT-IDENTICAL to the slab kernel
// (nh_kernels.h EdgeProfile) and the library's default (km <= 127).  A workgroup takes 16 consecutive columns of one of the two field
// pairs (crx / xfx on CX, cry / yfx on CY), a 16-lane row owns a column, a lane 8 consecutive interfaces; the two fields of a pair run
// side by side.  Row k (interface k) of the system of nh_utils.F90:1623-1660 is
//     y_k = (R_k - S_k y_(k-1)) / bet_k  downwards,   x_k = y_k - gam_k x_(k+1)  upwards,
// R_1 = xt1_top q(1) + q(2), R_k = 3 (q(k-1) + gk(k) q(k)), R_(km+1) = xt1_bot q(km) + q(km-1); S = 0, 1 .. 1, a_bot; the
// coefficients depend on dp0 only (the host's tables).  A lane runs its 8 rows from the value its neighbour hands it, round after
// round, until no hand-over moves (the recurrences forget: 1 / bet ~ gam ~ 0.27 a level, 3 - 4 rounds): then every lane holds the
// sequential sweep's bits (tridiag_rounds above).  Quotients by the host's correctly rounded 1 / bet and a Markstein correction = the
// values of `/`.  Every input is read once, every output written once, in 128-byte segments through LDS.
struct EdgeProfileLds {
  Grid g;
  int km;
  EdgeCoef ec;
  const double *rbet;     // device, km: 1 / ec.bet correctly rounded
  const double *q1, *q2;
  double *q1e, *q2e;
  int n2d;
  const double *q1_b, *q2_b;
  double *q1e_b, *q2e_b;
  int n2d_b;
  // The rows' coefficients as the host tabulates them (edge_rows() below: 6 x 128 doubles -- bet, 1 / bet, S, G, gk, the factor 3 of the
  // interior rows -- with the first, the bottom and the padded rows already in place).  tab = NULL: the wavefront builds them itself
  // with a dozen selects of doubles per row (a select of two doubles is 21 cycles of the SIMD, tools/lab/op_lab.hip: that set-up and
  // the three selects per row of the right-hand sides were half of the wavefront's instructions).  opt & 1: the two fields of a pair
  // run their rounds side by side (two independent chains: 5.15 instead of 6.3 cycles an operation).  opt & 2: the blocks of an XCD
  // are neighbours.  What the kernel waits for is memory, though (tools/lab/edge_lab.hip, C384 L127: 0.465 ms as round 5 left it and
  // with the table alone; 0.407 with the XCD ranges; 0.356 with three workgroups per CU as well -- tile_waves below; four spill)
  const double *tab = nullptr;
  int opt = 3;
  static constexpr int kTabRows = 16 * kFL, kTabDoubles = 6 * kTabRows;
  FV3_HD int nblk_a() const { return (n2d + kFC - 1) / kFC; }
  FV3_HD int nblocks() const { return nblk_a() + (n2d_b + kFC - 1) / kFC; }
  static constexpr int kIt = kFC * 128 / kNT;
  FV3_D void operator()(int bx, int, int, int tid, double *lds) const {
    double *B0 = lds, *B1 = lds + kFBuf;
    if (opt & 2) {   // neighbouring blocks share 128-byte lines: one contiguous range of blocks per XCD (xcd_block above)
      int by0 = 0;
      xcd_block(bx, by0, nblocks(), 1);
    }
    const bool second = bx >= nblk_a();
    const int blk = second ? bx - nblk_a() : bx;
    const ix_t ls = (ix_t)(second ? n2d_b : n2d);
    const double *f1 = second ? q1_b : q1, *f2 = second ? q2_b : q2;
    double *o1 = second ? q1e_b : q1e, *o2 = second ? q2e_b : q2e;
    const int c0g = blk * kFC;
    const int ncol = ((int)ls - c0g < kFC) ? (int)ls - c0g : kFC;
    {
      double v1[kIt], v2[kIt];
#ifndef FV3_HOST_EMU
#pragma unroll
#endif
      for (int it = 0; it < kIt; it++) {
        const int idx = tid + it * kNT, col = idx & (kFC - 1), k = idx >> 4;
        const ix_t o = (ix_t)(k < km ? k : km - 1) * ls + c0g + (col < ncol ? col : ncol - 1);
        v1[it] = f1[o];
        v2[it] = f2[o];
      }
#ifndef FV3_HOST_EMU
#pragma unroll
#endif
      for (int it = 0; it < kIt; it++) {
        const int idx = tid + it * kNT, col = idx & (kFC - 1), k = idx >> 4;
        B0[col * kFP + lds_lev(k)] = v1[it];
        B1[col * kFP + lds_lev(k)] = v2[it];
      }
    }
    FV3_SYNC();
#ifndef FV3_LAB_EDGE_OLD_PATH
    {   // (the library always hands the table over; the wavefront-built rows stay for the A / B of tools/lab/edge_lab.hip: compiled into
        // the same kernel they cost the table path 20 registers and nine spills under the three-wavefront budget)
      FV3_WAVE_FOR(wv) rows_from_table(wv * 4, B0, B1);
    }
#else
    if (tab) {
      FV3_WAVE_FOR(wv) rows_from_table(wv * 4, B0, B1);
    } else {
    const double xt2 = ec.gk_bot * (ec.gk_bot + 0.5) - ec.a_bot * ec.gam[km - 1];
    const double r_top = 1. / ec.bet_top, r_bot = 1. / xt2;
    FV3_WAVE_FOR(wv) {
      const int c0 = wv * 4;
      // the coefficients of the lane's rows (the same for every column)
      vd bet[kFL], rb[kFL], S[kFL], G[kFL], gkv[kFL];
      vb first[kFL], bot[kFL], mid[kFL];
      for (int q = 0; q < kFL; q++) {
        first[q] = vlevel_eq(q, 0);
        bot[q] = vlevel_eq(q, km);
        mid[q] = vlevel_lt(q, km) && !first[q];
        const vd bt = vrow_ld(ec.bet, q, km), rbt = vrow_ld(rbet, q, km);
        gkv[q] = vrow_ld(ec.gk, q, km);
        bet[q] = vsel(first[q], vd(ec.bet_top), vsel(bot[q], vd(xt2), vsel(mid[q], bt, vd(1.0))));
        rb[q] = vsel(first[q], vd(r_top), vsel(bot[q], vd(r_bot), vsel(mid[q], rbt, vd(1.0))));
        S[q] = vsel(mid[q], vd(1.0), vsel(bot[q], vd(ec.a_bot), vd(0.0)));
        G[q] = vsel(vlevel_lt(q, km), vrow_ld(ec.gam, q, km), vd(0.0));
      }
      vd x1[kFL], x2[kFL];
      for (int f = 0; f < 2; f++) {
        const double *B = f ? B1 : B0;
        vd a[kFL], R[kFL], y[kFL];
        vd *x = f ? x2 : x1;
        for (int q = 0; q < kFL; q++) a[q] = vlds_ld(B, c0, q);
        const vd a_up1 = row_shr<1>(a[kFL - 1], 0.0), a_up2 = row_shr<1>(a[kFL - 2], 0.0), a_dn = row_shl<1>(a[0], 0.0);
        for (int q = 0; q < kFL; q++) {
          const vd am1 = q > 0 ? a[q - 1] : a_up1, am2 = q > 1 ? a[q - 2] : (q == 1 ? a_up1 : a_up2);
          const vd ap1 = q < kFL - 1 ? a[q + 1] : a_dn;
          const vd r_first = ec.xt1_top * a[q] + ap1;                   // :1631
          const vd r_mid = 3. * (am1 + gkv[q] * a[q]);                   // :1636
          const vd r_bot_ = ec.xt1_bot * am1 + am2;                      // :1648 (q(km), q(km-1))
          R[q] = vsel(first[q], r_first, vsel(bot[q], r_bot_, vsel(mid[q], r_mid, vd(0.0))));
        }
        {
          vd yin(0.0);
          for (int rnd = 0; rnd < 16; rnd++) {
            vd v = yin;
            for (int q = 0; q < kFL; q++) {
              v = vdiv_r(R[q] - S[q] * v, bet[q], rb[q]);
              y[q] = v;
            }
            const vd ynew = row_shr<1>(v, 0.0);
            const bool moved = vany_ne(ynew, yin);
            yin = ynew;
            if (!moved) break;
          }
        }
        {
          vd xin(0.0);
          for (int rnd = 0; rnd < 16; rnd++) {
            vd xn = xin;
            for (int q = kFL - 1; q >= 0; q--) {
              xn = y[q] - G[q] * xn;
              x[q] = xn;
            }
            const vd xnew = row_shl<1>(xn, 0.0);
            const bool moved = vany_ne(xnew, xin);
            xin = xnew;
            if (!moved) break;
          }
        }
      }
      for (int q = 0; q < kFL; q++) {       // the wavefront's own columns: no other wavefront reads them
        vlds_st(B0, c0, q, x1[q]);
        vlds_st(B1, c0, q, x2[q]);
      }
    }
    }
#endif
    FV3_SYNC();
    for (int idx = tid; idx < kFC * 128; idx += kNT) {
      const int col = idx & (kFC - 1), k = idx >> 4;
      if (k <= km && col < ncol) {
        const ix_t o = (ix_t)k * ls + c0g + col;
        o1[o] = B0[col * kFP + lds_lev(k)];
        o2[o] = B1[col * kFP + lds_lev(k)];
      }
    }
  }
  // right-hand sides of the lane's rows: the interior expression everywhere (3 (q(k-1) + gk q(k)), the factor 0 on the padded rows), the
  // first row (lane 0, q = 0) and the bottom row (row km: ONE q for the whole wavefront, a scalar branch) put in place by a select each
  FV3_D void rhs(const double *B, int c0, const vd *gkv, const vd *c3, vd *R) const {
    vd a[kFL];
    for (int q = 0; q < kFL; q++) a[q] = vlds_ld(B, c0, q);
    const vd a_up1 = row_shr<1>(a[kFL - 1], 0.0), a_up2 = row_shr<1>(a[kFL - 2], 0.0);
    const int qb = km & (kFL - 1);
#ifndef FV3_HOST_EMU
#pragma unroll
#endif
    for (int q = 0; q < kFL; q++) {
      const vd am1 = q > 0 ? a[q - 1] : a_up1;
      vd r = c3[q] * (am1 + gkv[q] * a[q]);                                                   // :1636
      if (q == 0) r = vsel(vlevel_eq(0, 0), ec.xt1_top * a[0] + a[1], r);                     // :1631
      if (q == qb) {
        const vd am2 = q > 1 ? a[q - 2] : (q == 1 ? a_up1 : a_up2);
        r = vsel(vlevel_eq(q, km), ec.xt1_bot * am1 + am2, r);                                // :1648 (q(km), q(km-1))
      }
      R[q] = r;
    }
  }
  FV3_D void rows_from_table(int c0, double *B0, double *B1) const {
    vd bet[kFL], rb[kFL], S[kFL], G[kFL];
    vd R1[kFL], R2[kFL];
    {
      vd gkv[kFL], c3[kFL];
      for (int q = 0; q < kFL; q++) {
        bet[q] = vrow_ld(tab, q, kTabRows);
        rb[q] = vrow_ld(tab + kTabRows, q, kTabRows);
        S[q] = vrow_ld(tab + 2 * kTabRows, q, kTabRows);
        G[q] = vrow_ld(tab + 3 * kTabRows, q, kTabRows);
        gkv[q] = vrow_ld(tab + 4 * kTabRows, q, kTabRows);
        c3[q] = vrow_ld(tab + 5 * kTabRows, q, kTabRows);
      }
      rhs(B0, c0, gkv, c3, R1);
      rhs(B1, c0, gkv, c3, R2);
    }
    vd y1[kFL], y2[kFL];
#ifdef FV3_LAB_EDGE_SIDE
    if (FV3_LAB_EDGE_SIDE) {
#else
    if (opt & 1) {
#endif
      vd in1(0.0), in2(0.0);
      for (int rnd = 0; rnd < 16; rnd++) {
        vd v1 = in1, v2 = in2;
#ifndef FV3_HOST_EMU
#pragma unroll
#endif
        for (int q = 0; q < kFL; q++) {
          v1 = vdiv_r(R1[q] - S[q] * v1, bet[q], rb[q]);
          v2 = vdiv_r(R2[q] - S[q] * v2, bet[q], rb[q]);
          y1[q] = v1;
          y2[q] = v2;
        }
        const vd n1 = row_shr<1>(v1, 0.0), n2 = row_shr<1>(v2, 0.0);
        const bool moved = vany_ne(n1, in1) || vany_ne(n2, in2);     // a round more than a field needs leaves it as it is
        in1 = n1;
        in2 = n2;
        if (!moved) break;
      }
      in1 = vd(0.0);
      in2 = vd(0.0);
      for (int rnd = 0; rnd < 16; rnd++) {
        vd v1 = in1, v2 = in2;
#ifndef FV3_HOST_EMU
#pragma unroll
#endif
        for (int q = kFL - 1; q >= 0; q--) {
          v1 = y1[q] - G[q] * v1;
          v2 = y2[q] - G[q] * v2;
          R1[q] = v1;
          R2[q] = v2;
        }
        const vd n1 = row_shl<1>(v1, 0.0), n2 = row_shl<1>(v2, 0.0);
        const bool moved = vany_ne(n1, in1) || vany_ne(n2, in2);
        in1 = n1;
        in2 = n2;
        if (!moved) break;
      }
    } else {
      for (int f = 0; f < 2; f++) {
        vd *R = f ? R2 : R1, *y = f ? y2 : y1;
        vd in(0.0);
        for (int rnd = 0; rnd < 16; rnd++) {
          vd v = in;
          for (int q = 0; q < kFL; q++) {
            v = vdiv_r(R[q] - S[q] * v, bet[q], rb[q]);
            y[q] = v;
          }
          const vd nw = row_shr<1>(v, 0.0);
          const bool moved = vany_ne(nw, in);
          in = nw;
          if (!moved) break;
        }
        in = vd(0.0);
        for (int rnd = 0; rnd < 16; rnd++) {
          vd v = in;
          for (int q = kFL - 1; q >= 0; q--) {
            v = y[q] - G[q] * v;
            R[q] = v;
          }
          const vd nw = row_shl<1>(v, 0.0);
          const bool moved = vany_ne(nw, in);
          in = nw;
          if (!moved) break;
        }
      }
    }
    for (int q = 0; q < kFL; q++) {       // the wavefront's own columns: no other wavefront reads them
      vlds_st(B0, c0, q, R1[q]);
      vlds_st(B1, c0, q, R2[q]);
    }
  }
};

template <>
struct tile_waves<EdgeProfileLds> { static constexpr int value = 3; };

// The table EdgeProfileLds::tab points at (host; t: 6 x 128 doubles), from the coefficients fv3_set_dp_ref has just computed (host
// copies gk, bet, gam, rbet of km entries each).  Row k is interface k of the system of nh_utils.F90:1623-1660.
inline void edge_rows(double *t, int km, const double *gk, const double *bet, const double *gam, const double *rbet, double bet_top,
                      double a_bot, double gk_bot) {
  constexpr int n = EdgeProfileLds::kTabRows;
  const double xt2 = gk_bot * (gk_bot + 0.5) - a_bot * gam[km - 1];
  for (int k = 0; k < n; k++) {
    double b = 1., r = 1., s = 0., g = 0., gkk = 0., c3 = 0.;
    if (k == 0) { b = bet_top; r = 1. / bet_top; g = gam[0]; }
    else if (k < km) { b = bet[k]; r = rbet[k]; s = 1.; g = gam[k]; gkk = gk[k]; c3 = 3.; }
    else if (k == km) { b = xt2; r = 1. / xt2; s = a_bot; }
    t[k] = b; t[n + k] = r; t[2 * n + k] = s; t[3 * n + k] = g; t[4 * n + k] = gkk; t[5 * n + k] = c3;
  }
}

// CG = true: Riem_Solver_c on (is-1:ie+1, js-1:je+1); false: Riem_Solver3 on the compute domain
//
// BIT-IDENTICAL to the slab kernels (nh_kernels.h sim_column) (EX: kept as a template argument of the instantiations, always true).
// Everything pointwise in k already was the parity kernel's expression; what differed were the recurrences in k.  They now run in the
// reference's own order:
//   * the sums (hydrostatic pressure downwards, pe2 downwards, p1 and the heights upwards): a lane runs its 8 levels from the value
//     its neighbour hands it, all lanes at once, round after round; after round r the first r lanes hold the sequential values, so
//     km / 8 + 1 rounds ARE the sequential sweep (a sum forgets nothing, so no early exit) -- 10 instructions per round;
//   * the pp system (:1302-1332): the same rounds with an early exit when no hand-over moves any more (remap_fast.h spline(): the
//     elimination of a diagonally dominant system forgets, d gam_k / d gam_(k-1) ~ 0.1: 3 - 4 rounds);
//   * the w system (:1335-1361) is stiff (|aa| / dm = 2 (c dt / dz)^2 ~ 1e3 .. 1e6: its elimination forgets nothing within 128 levels)
//     and has two divisions per level: 16 rounds would cost 4 x what the sweep costs.  Its coefficients go to LDS (the transposition
//     buffers, free by then) and ONE wavefront of the workgroup runs the 16 columns of the workgroup on 16 lanes, a lane per column,
//     with the parity kernel's own statements (rcp_rn / div_rn); the other wavefronts wait at the barrier -- the second workgroup of
//     the CU has the SIMDs meanwhile.
template <bool CG, bool EX = true, bool SIM = false, bool MOIST = false>
struct RiemFast {
  static_assert(EX, "the tolerance mode (blocked parallel scans, not bit-identical: whole steps left 1e-12) was removed in round 5");
  static_assert(!SIM || !CG, "SIM_solver: the D grid's Riem_Solver3");
  Grid g;
  int km;
  double dt;
  NhConsts cn;
  // inputs (names of RiemSolver3 / RiemSolverC): zs = zs / hs; wq = w / w3; zl = zh / gz (updated in place)
  const double *zs, *pt, *delp, *ws;
  double *wq, *zl;
  // outputs: D grid: delz, ppe, pk3 (+ pe, pk, peln on the last call); C grid: pef
  double *delz, *ppe, *pk3, *pe, *pk, *peln, *pef;
  int use_logp, last_call, fp_out;
  // MOIST: q_con (use_cond: the condensates leave the hydrostatic pressure of pm2, nh_core.F90:113-131, :145-154 / nh_utils.F90:383-396,
  // :413-438) and cappa (moist_kappa: gm2, cp2 per cell; on the C grid only together with q_con), A x km or null
  const double *qcon = nullptr, *cappa = nullptr;
  int probe = 0;   // timing probe (tools/riem_time.py, FV3_MI355X_RIEM_PROBE): 1 one round per sum, 2 no w pass, 4 one round of the pp system -- WRONG results
  int opt = 7;     // A / B switches of tools/lab/riem_lab.hip (all on in the library): 1 the pass wavefront chosen by SIMD, 2 the sweep in
                   // alternating register sets, 4 XCD-contiguous column blocks
  long long *trace = nullptr;   // tools/lab (-DFV3_LAB_TRACE): clock64() of every wavefront at the barriers of block trace_blk
  int trace_blk = -1;
  // LDS: the four transposition buffers + the wavefronts' hardware ids (pass_wave)
  static constexpr size_t kLdsDoubles = (size_t)kFNBuf * kFBuf + 8;

  FV3_HD int i_first() const { return CG ? g.is - 1 : g.is; }
  FV3_HD int ncols_row() const { return CG ? g.nx + 2 : g.nx; }
  FV3_HD int nrows() const { return CG ? g.ny + 2 : g.ny; }
  FV3_HD int nblocks_x() const { return (ncols_row() + kFC - 1) / kFC; }

  // one level-major A-layout field -> LDS [column][level]; lev levels (<= 128); fill for what does not exist.  All loads of the
  // workgroup's share are issued before the first LDS store (clamped addresses: no branch between them), so the four fields' 32
  // loads per thread are in flight together instead of one HBM round trip after the other.
  static constexpr int kIt = kFC * 128 / kNT;
  FV3_D void stage_load(double *v, const double *f, ix_t o0, int ncol, int lev, int tid) const {
    const ix_t nA = g.nA();
#ifndef FV3_HOST_EMU
#pragma unroll
#endif
    for (int it = 0; it < kIt; it++) {
      const int idx = tid + it * kNT, col = idx & (kFC - 1), k = idx >> 4;
      v[it] = f[(ix_t)(k < lev ? k : lev - 1) * nA + o0 + (col < ncol ? col : ncol - 1)];
    }
  }
  FV3_D void stage_store(double *buf, const double *v, int ncol, int lev, double fill, int tid) const {
#ifndef FV3_HOST_EMU
#pragma unroll
#endif
    for (int it = 0; it < kIt; it++) {
      const int idx = tid + it * kNT, col = idx & (kFC - 1), k = idx >> 4;
      buf[col * kFP + lds_lev(k)] = (k < lev && col < ncol) ? v[it] : fill;
    }
  }
  template <class Addr>
  FV3_D void stage_out(const double *buf, double *f, int ncol, int lev, int tid, const Addr &addr) const {
    for (int idx = tid; idx < kFC * 128; idx += kNT) {
      const int col = idx & (kFC - 1), k = idx >> 4;
      if (k < lev && col < ncol) f[addr(col, k)] = buf[col * kFP + lds_lev(k)];
    }
  }

  // EX: the w system of one column (SIM1_solver nh_utils.F90:1335-1361) with the parity kernel's statements (nh_kernels.h sim_column
  // passes C and D), its rows in LDS: A = aa at the top interface of layer k (k = km: p1 of the bottom layer), D = dm, R = right-hand
  // side.  On exit R = w2, A = gam.  Chunks of 8 levels are loaded ahead of the recurrence that consumes them.
  FV3_D void w_column(double *A, double *D, double *R) const {
    double bet = 1., rbet = 1., y = 0.;
    const int nch = (km + kFL - 1) / kFL;
    double an[kFL + 1], dn[kFL], rn[kFL];
    for (int u = 0; u < kFL; u++) { an[u] = A[u]; dn[u] = D[u]; rn[u] = R[u]; }
    an[kFL] = A[kFS];                                        // the row below the chunk's last: the next chunk's first word
    for (int ch = 0; ch < nch; ch++) {
      double ac[kFL + 1], dc[kFL], rc[kFL];
      for (int u = 0; u <= kFL; u++) ac[u] = an[u];
      for (int u = 0; u < kFL; u++) { dc[u] = dn[u]; rc[u] = rn[u]; }
      if (ch + 1 < nch) {
        const int o = (ch + 1) * kFS;
        for (int u = 0; u < kFL; u++) { an[u] = A[o + u]; dn[u] = D[o + u]; rn[u] = R[o + u]; }
        an[kFL] = A[o + kFS];
      }
      double gv[kFL], yv[kFL];
      // (the rows of the last chunk beyond km are padding -- aa = 0, dm = 1, rhs = 0 from the pointwise part, so bet = 1 there and
      // nothing is selected per level: the recurrence is 17 dependent instructions a level and nothing else)
#ifndef FV3_HOST_EMU
#pragma unroll
#endif
      for (int u = 0; u < kFL; u++) {
        const double a = ac[u], low = ac[u + 1];
        const double gam = div_rn(a, bet, rbet);
        bet = dc[u] - (a + low + a * gam);
        rbet = rcp_rn(bet);
        y = div_rn(rc[u] - a * y, bet, rbet);
        gv[u] = gam;
        yv[u] = y;
      }
      const int o = ch * kFS;
      for (int u = 0; u < kFL; u++) {
        A[o + u] = gv[u];
        R[o + u] = yv[u];
      }
    }
    // back substitution: w2(k) = w2(k) - gam(k+1) w2(k+1), k = km-1 .. 1
    double wn = 0.;
    for (int ch = nch - 1; ch >= 0; ch--) {
      const int o = ch * kFS;
      double yc[kFL], gc[kFL + 1];
      for (int u = 0; u < kFL; u++) yc[u] = R[o + u];
      for (int u = 1; u < kFL; u++) gc[u] = A[o + u];
      gc[kFL] = A[o + kFS];                                  // gam of the next chunk's first row
#ifndef FV3_HOST_EMU
#pragma unroll
#endif
      for (int u = kFL - 1; u >= 0; u--) {
        const int k = ch * kFL + u;
        const double v = k == km - 1 ? yc[u] : yc[u] - gc[u + 1] * wn;
        wn = k <= km - 1 ? v : wn;
        yc[u] = wn;
      }
      for (int u = 0; u < kFL; u++)
        if (ch * kFL + u < km) R[o + u] = yc[u];
    }
  }
  // The same sweep as the library runs it (opt & 2; w_column above is kept for the A / B of tools/lab/riem_lab.hip).  What the pass costs
  // is ISSUE: an f64 operation takes 5.15 cycles of its SIMD whatever the active lanes (tools/lab/op_lab.hip), a level of the forward
  // sweep is 17 of them, one a reciprocal (16 cycles) -- 100 cycles a level at best, 16 lanes busy, three wavefronts waiting.  So:
  //   * the chunks alternate between two register sets (the copies "next -> current" were a quarter of the sweep's instructions);
  //   * the back substitution reads its chunks one ahead, stores every row (the padded ones hold 0) and branches nowhere: 100 -> 25
  //     cycles a level (it was 40 % of the pass: eight branches around eight stores per chunk, every load waited for on the spot);
  //   * a padded chunk beyond the last is swept too when the number of chunks is odd (aa = 0, dm = 1, rhs = 0 there).
  FV3_D void w_chunk(const double *a, const double *dc, const double *rc, double &bet, double &rbet, double &y, double *gv, double *yv) const {
#ifndef FV3_HOST_EMU
#pragma unroll
#endif
    for (int u = 0; u < kFL; u++) {
      const double av = a[u], low = a[u + 1];
      const double gam = div_rn(av, bet, rbet);
      bet = dc[u] - (av + low + av * gam);
      rbet = rcp_rn(bet);
      y = div_rn(rc[u] - av * y, bet, rbet);
      gv[u] = gam;
      yv[u] = y;
    }
  }
  FV3_D void w_column2(double *A, double *D, double *R) const {
    double bet = 1., rbet = 1., y = 0.;
    const int nch = (km + kFL - 1) / kFL, nch2 = (nch + 1) & ~1;
    double a0[kFL + 1], d0[kFL], r0[kFL], a1[kFL + 1], d1[kFL], r1[kFL];
    for (int u = 0; u < kFL; u++) { a0[u] = A[u]; d0[u] = D[u]; r0[u] = R[u]; }
    a0[kFL] = A[kFS];
    for (int ch = 0; ch < nch2; ch += 2) {
      {
        const int o = (ch + 1) * kFS;
        for (int u = 0; u < kFL; u++) { a1[u] = A[o + u]; d1[u] = D[o + u]; r1[u] = R[o + u]; }
        a1[kFL] = A[o + kFS];
      }
      double gv[kFL], yv[kFL];
      w_chunk(a0, d0, r0, bet, rbet, y, gv, yv);
      {
        const int o = ch * kFS;
        for (int u = 0; u < kFL; u++) { A[o + u] = gv[u]; R[o + u] = yv[u]; }
      }
      if (ch + 2 < nch2) {
        const int o = (ch + 2) * kFS;
        for (int u = 0; u < kFL; u++) { a0[u] = A[o + u]; d0[u] = D[o + u]; r0[u] = R[o + u]; }
        a0[kFL] = A[o + kFS];
      }
      w_chunk(a1, d1, r1, bet, rbet, y, gv, yv);
      {
        const int o = (ch + 1) * kFS;
        for (int u = 0; u < kFL; u++) { A[o + u] = gv[u]; R[o + u] = yv[u]; }
      }
    }
    // back substitution: w2(k) = w2(k) - gam(k+1) w2(k+1), k = km-1 .. 1; rows k >= km keep 0.  Only the chunk that holds row km - 1
    // selects per row (a select of doubles in the recurrence costs more than the recurrence: 21 cycles against 12); the chunks above
    // it hold real rows only.
    double wn = 0.;
    double yn[kFL], gn[kFL + 1];
    {
      const int o = (nch - 1) * kFS;
      for (int u = 0; u < kFL; u++) yn[u] = R[o + u];
      for (int u = 1; u < kFL; u++) gn[u] = A[o + u];
      gn[kFL] = A[o + kFS];                                // gam of the next chunk's first row
    }
    for (int ch = nch - 1; ch >= 0; ch--) {
      double yc[kFL], gc[kFL + 1];
      for (int u = 0; u < kFL; u++) { yc[u] = yn[u]; gc[u + 1] = gn[u + 1]; }
      {   // the chunk above, requested before this one is swept (the topmost chunk asks for itself again)
        const int o = (ch > 0 ? ch - 1 : 0) * kFS;
        for (int u = 0; u < kFL; u++) yn[u] = R[o + u];
        for (int u = 1; u < kFL; u++) gn[u] = A[o + u];
        gn[kFL] = A[o + kFS];
      }
      if (ch == nch - 1) {
#ifndef FV3_HOST_EMU
#pragma unroll
#endif
        for (int u = kFL - 1; u >= 0; u--) {
          const int k = ch * kFL + u;
          const double v = k == km - 1 ? yc[u] : yc[u] - gc[u + 1] * wn;
          wn = k <= km - 1 ? v : wn;
          yc[u] = wn;
        }
      } else {
#ifndef FV3_HOST_EMU
#pragma unroll
#endif
        for (int u = kFL - 1; u >= 0; u--) {
          wn = yc[u] - gc[u + 1] * wn;
          yc[u] = wn;
        }
      }
      const int o = ch * kFS;
      for (int u = 0; u < kFL; u++) R[o + u] = yc[u];
    }
  }
  // one wavefront of the workgroup (`wave`, so that the workgroups of a CU use different SIMDs for it) runs the 16 columns, a lane each
  FV3_D void w_columns(double *A, double *D, double *R, int wave, int tid) const {
#ifdef FV3_HOST_EMU
    (void)wave; (void)tid;
    for (int col = 0; col < kFC; col++) w_column(A + col * kFP, D + col * kFP, R + col * kFP);
#else
    if ((tid >> 6) == wave && (tid & 63) < kFC) {
      const int col = tid & 63;
      if (opt & 2) w_column2(A + col * kFP, D + col * kFP, R + col * kFP);
      else w_column(A + col * kFP, D + col * kFP, R + col * kFP);
    }
#endif
  }

  // (the barriers order LDS only, FV3_SYNC_LDS: every field is read before the first and written after the last exchange through LDS,
  // by the threads of this workgroup alone, so the output stores of one phase drain behind the transposition of the next)
  // tools/lab (-DFV3_LAB_TRACE): where a block's time goes -- clock64() of every wavefront on either side of the barriers
#ifdef FV3_LAB_TRACE
#define FV3_TRACE(n) do { if (trace && by * nblocks_x() + bx == trace_blk && (tid & 63) == 0) trace[(tid >> 6) * 16 + (n)] = clock64(); } while (0)
#else
#define FV3_TRACE(n) ((void)0)
#endif
  FV3_D void operator()(int bx, int by, int, int tid, double *lds) const {
    double *B0 = lds, *B1 = lds + kFBuf, *B2 = lds + 2 * kFBuf, *B3 = lds + 3 * kFBuf;
    if (opt & 4) xcd_block(bx, by, nblocks_x(), nrows());
    FV3_TRACE(15);
    const int i0 = i_first() + bx * kFC, j = (CG ? g.js - 1 : g.js) + by;
    const int ncol = (ncols_row() - bx * kFC < kFC) ? ncols_row() - bx * kFC : kFC;
    const ix_t nA = g.nA(), nCC = g.nCC();
    const ix_t o0 = (ix_t)g.iA(i0, j);
    const double rgrav = 1. / cn.grav, rgas = cn.rdgas, gm2 = 1. / (1. - cn.akap), cp2 = cn.akap;
    // SIM: SIM_solver (nh_utils.F90:1396-1537, a_imp < 1: the off-centred form) -- t1g with alpha dt, the explicit part wk of the w
    // equation, the blend of pe2 with pp at the end; everything else is SIM1_solver
    const double alpha = cn.a_imp, beta = 1. - alpha, ra = 1. / alpha, t2 = beta / alpha;
    const double t1g = SIM ? 2. * ((alpha * dt) * (alpha * dt)) : 2. * dt * dt, rdt = 1. / dt;
    constexpr double r3 = 1. / 3.;
    vd dmr[kWvState][kFL], ptv[kWvState][kFL], w1[kWvState][kFL], zv[kWvState][kFL + 1];
    vd keep_pm2[kWvState][kFL], keep_grat[kWvState][kFL];
    vd keep_pem[kWvState][kFL + 1], keep_ppt[kWvState][SIM ? kFL + 1 : 1];
    const int nrounds = (probe & 1) ? 1 : km / kFL + 1;   // EX: rounds after which the hand-overs of a sequential sweep over km levels are the sweep's own
    // ---- inputs: the four fields at once (their loads are in flight together), one barrier ----
    {
      double v0[kIt], v1[kIt], v2[kIt], v3[kIt];
      stage_load(v0, delp, o0, ncol, km, tid);
      stage_load(v1, pt, o0, ncol, km, tid);
      stage_load(v2, wq, o0, ncol, km, tid);
      stage_load(v3, zl, o0, ncol, km + 1, tid);
      stage_store(B0, v0, ncol, km, 1.0, tid);
      stage_store(B1, v1, ncol, km, 300.0, tid);
      stage_store(B2, v2, ncol, km, 0.0, tid);
      stage_store(B3, v3, ncol, km + 1, 0.0, tid);
      // interface heights beyond km+1 continue downwards so that the padded layers stay regular (dz < 0)
      for (int idx = tid; idx < kFC * 128; idx += kNT) {
        const int col = idx & (kFC - 1), k = idx >> 4;
        if (k > km || col >= ncol) B3[col * kFP + lds_lev(k)] = -1.0e4 - 10. * (double)k;
      }
    }
    int pass_wave = (bx + by) & 3;
#ifndef FV3_HOST_EMU
    // The wavefront that runs the w pass.  Two workgroups share a CU and a pass keeps its SIMD issuing for ~15 000 cycles while the other
    // three wait: the two passes must not meet on one SIMD.  Every wavefront posts the SIMD it runs on (HW_ID[5:4]), the first one also
    // its wave slot (HW_ID[3:0]: co-resident workgroups hold different slots); the pass goes to the wavefront on SIMD (slot mod 4), or
    // to wavefront (slot mod 4) should the four not sit on four SIMDs.
    int *hw = reinterpret_cast<int *>(lds + kFNBuf * kFBuf);
    if ((opt & 1) && (tid & 63) == 0) {
      hw[tid >> 6] = (int)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);
      if (tid == 0) hw[4] = (int)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);
    }
#endif
    FV3_TRACE(0);
    FV3_SYNC_LDS();
    FV3_TRACE(1);
#ifndef FV3_HOST_EMU
    if (opt & 1) {
      const int target = hw[4] & 3;
      pass_wave = target;
      for (int w_ = 3; w_ >= 0; w_--)
        if (hw[w_] == target) pass_wave = w_;
    }
#endif
    FV3_WAVE_FOR(wv) {
      const int s = FV3_WVI(wv), c0 = wv * 4;
      for (int q = 0; q < kFL; q++) {
        dmr[s][q] = vlds_ld(B0, c0, q);
        ptv[s][q] = vlds_ld(B1, c0, q);
        w1[s][q] = vlds_ld(B2, c0, q);
        zv[s][q] = vlds_ld(B3, c0, q);
      }
      zv[s][kFL] = row_shl<1>(zv[s][0], -2.0e4);   // the interface below the lane's last layer (lane 15: a padded layer)
    }
    FV3_SYNC_LDS();
    const bool has_qc = MOIST && qcon != nullptr, has_cappa = MOIST && cappa != nullptr && (!CG || qcon != nullptr);
    vd qcv[kWvState][MOIST ? kFL : 1], cpv[kWvState][MOIST ? kFL : 1];
    if constexpr (MOIST) {
      {
        double v0[kIt], v1[kIt];
        stage_load(v0, has_qc ? qcon : delp, o0, ncol, km, tid);
        stage_load(v1, has_cappa ? cappa : delp, o0, ncol, km, tid);
        stage_store(B0, v0, ncol, km, 0.0, tid);
        stage_store(B1, v1, ncol, km, cn.akap, tid);
      }
      FV3_SYNC_LDS();
      FV3_WAVE_FOR(wv) {
        const int s = FV3_WVI(wv), c0 = wv * 4;
        for (int q = 0; q < kFL; q++) {
          qcv[s][q] = has_qc ? vlds_ld(B0, c0, q) : vd(0.0);
          cpv[s][q] = has_cappa ? vlds_ld(B1, c0, q) : vd(cn.akap);
        }
      }
      FV3_SYNC_LDS();
    }
    // ---- the column: everything below is per wavefront, no barrier until the outputs ----
    FV3_WAVE_FOR(wv) {
      const int s = FV3_WVI(wv), c0 = wv * 4;
      vd dm[kFL], dz[kFL], pm2[kFL], pei[kFL], grat[kFL], bb[kFL], lnp[kFL + 1], X[kFL], pemv[kFL + 1];
      vb real[kFL], last[kFL];
      for (int q = 0; q < kFL; q++) {
        real[q] = vlevel_lt(q, km);
        last[q] = vlevel_eq(q, km - 1);
      }
      {  // hydrostatic pressure at the interfaces: pem(k+1) = pem(k) + delp(k) (nh_core.F90:132-141 / nh_utils.F90:404-441), the
         // lane's 8 levels in the reference's order on top of the scanned sum of the lanes above
        {
          vd in(cn.ptop);
          for (int rnd = 0; rnd < nrounds; rnd++) {
            vd run = in;
            for (int q = 0; q < kFL; q++) {
              pemv[q] = run;
              run = run + dmr[s][q];
            }
            pemv[kFL] = run;
            in = row_shr<1>(run, cn.ptop);
          }
        }
      }
      // hydrostatic pressure functions, perturbation pressure (nh_utils.F90:1297-1300; nh_core.F90:140-159 / nh_utils.F90:440)
      if (!CG) {
        for (int q = 0; q < kFL; q++) lnp[q] = vlog(pemv[q]);
        lnp[kFL] = row_shl<1>(lnp[0], 0.0);
      }
      vd gm2q[MOIST ? kFL : 1];
      vd pegv[MOIST ? kFL + 1 : 1];
      if constexpr (MOIST) {
        for (int q = 0; q < kFL; q++) gm2q[q] = has_cappa ? vdivq(vd(1.0), 1. - cpv[s][q]) : vd(gm2);
        if (has_qc) {     // peg(k+1) = peg(k) + delp (1 - q_con) from ptop, in the reference's order
          vd in(cn.ptop);
          for (int rnd = 0; rnd < nrounds; rnd++) {
            vd run = in;
            for (int q = 0; q < kFL; q++) {
              pegv[q] = run;
              run = run + dmr[s][q] * (1. - qcv[s][q]);
            }
            pegv[kFL] = run;
            in = row_shr<1>(run, cn.ptop);
          }
        }
      }
      vd lng[MOIST ? kFL + 1 : 1];
      if constexpr (MOIST) {
        if (has_qc && !CG) {
          for (int q = 0; q < kFL; q++) lng[q] = vlog(pegv[q]);
          lng[kFL] = row_shl<1>(lng[0], 0.0);
        }
      }
      for (int q = 0; q < kFL; q++) {
        const vd d = dmr[s][q];
        if (CG)
          pm2[q] = vdivq(d, vlog(vdivq(pemv[q + 1], pemv[q])));
        else
          pm2[q] = vdivq(d, lnp[q + 1] - lnp[q]);
        if constexpr (MOIST) {
          if (has_qc) {     // excluding the contribution from the condensates
            if (CG)
              pm2[q] = vdivq(pegv[q + 1] - pegv[q], vlog(vdivq(pegv[q + 1], pegv[q])));
            else
              pm2[q] = vdivq(pegv[q + 1] - pegv[q], lng[q + 1] - lng[q]);
          }
        }
        dm[q] = d * rgrav;
        dz[q] = zv[s][q + 1] - zv[s][q];
        if constexpr (MOIST)
          pei[q] = vexp(gm2q[q] * vlog(vdivq(-dm[q], dz[q]) * rgas * ptv[s][q])) - pm2[q];
        else
          pei[q] = vexp(gm2 * vlog(vdivq(-dm[q], dz[q]) * rgas * ptv[s][q])) - pm2[q];
      }
      // ---- pp: forward / backward elimination of nh_utils.F90:1302-1332 as one tridiagonal system; X(k) = pp(k+1) ----
      {
        vd a[kFL], c[kFL], d[kFL];
        const vd dm_nx = row_shl<1>(dm[0], 1.0), pe_nx = row_shl<1>(pei[0], 0.0);
        for (int q = 0; q < kFL; q++) {
          const vd dmn = (q < kFL - 1) ? dm[q + 1] : dm_nx, pen = (q < kFL - 1) ? pei[q + 1] : pe_nx;
          const vd gr = vdivq(dm[q], dmn);
          grat[q] = vsel(last[q], vd(0.0), gr);
          bb[q] = vsel(last[q], vd(2.0), 2. * (1. + gr));
          const vd dd = vsel(last[q], 3. * pei[q], 3. * (pei[q] + gr * pen));
          a[q] = vsel(real[q] && !vlevel_eq(q, 0), vd(1.0), vd(0.0));
          c[q] = vsel(real[q], grat[q], vd(0.0));
          d[q] = vsel(real[q], dd, vd(0.0));
          bb[q] = vsel(real[q], bb[q], vd(1.0));
        }
        tridiag_rounds(a, bb, c, d, X, (probe & 4) ? 1 : 16);
      }
      // ---- w: nh_utils.F90:1335-1361 ----
      {
        vd a[kFL], b[kFL], c[kFL], d[kFL], aat[kFL];
        const vd dz_pv = row_shr<1>(dz[kFL - 1], 1.0), X_pv = row_shr<1>(X[kFL - 1], 0.0);
        for (int q = 0; q < kFL; q++) {   // aa at the top interface of the layer (0 at the model top)
          const vd dzp = (q > 0) ? dz[q - 1] : dz_pv;
          vd aa;
          if constexpr (MOIST) {
            const vd gm2p = (q > 0) ? gm2q[q - 1] : row_shr<1>(gm2q[kFL - 1], gm2);
            aa = vdivq(t1g * 0.5 * (gm2p + gm2q[q]), dzp + dz[q]) * pemv[q];
          } else {
            aa = vdivq(vd(t1g * 0.5 * (gm2 + gm2)), dzp + dz[q]) * pemv[q];
          }
          aat[q] = vsel(real[q] && !vlevel_eq(q, 0), aa, vd(0.0));
        }
        const vd aat_nx = row_shl<1>(aat[0], 0.0);
        const vd wsv = vcol_ld(ws, CG ? (long)o0 : (long)g.iCC(i0, j), c0, ncol);
        for (int q = 0; q < kFL; q++) {
          const vd aab = (q < kFL - 1) ? aat[q + 1] : aat_nx;
          vd p1c;                                                        // bottom layer only (:1349)
          if constexpr (MOIST)
            p1c = vdivq(t1g * gm2q[q], dz[q]) * pemv[q + 1];
          else
            p1c = vdivq(vd(t1g * gm2), dz[q]) * pemv[q + 1];
          const vd ppt = (q > 0) ? X[q - 1] : X_pv;                // pp at the top interface of the layer
          const vd low = vsel(last[q], p1c, aab);
          a[q] = aat[q];
          b[q] = vsel(real[q], dm[q] - (aat[q] + low), vd(1.0));
          c[q] = vsel(real[q] && !last[q], aab, vd(0.0));
          const vd rhs = dm[q] * w1[s][q] + dt * (X[q] - ppt);
          if constexpr (SIM) {   // wk(k) = t2 aa(k) (w1(k-1) - w1(k)) at the layer's top and bottom interfaces (:1465-1466)
            const vd w1p = (q > 0) ? w1[s][q - 1] : row_shr<1>(w1[s][kFL - 1], 0.0);
            const vd w1n = (q < kFL - 1) ? w1[s][q + 1] : row_shl<1>(w1[s][0], 0.0);
            const vd wkt = t2 * aat[q] * (w1p - w1[s][q]), wkb = t2 * aab * (w1[s][q] - w1n);
            d[q] = vsel(real[q], vsel(last[q], rhs - wkt + p1c * (t2 * w1[s][q] - ra * wsv), rhs + wkb - wkt), vd(0.0));
          } else {
            d[q] = vsel(real[q], vsel(last[q], rhs - p1c * wsv, rhs), vd(0.0));
          }
        }
        {
          // coefficients of the workgroup's 16 columns -> LDS: aa at the top interface of every layer (the model's bottom interface
          // carries p1 of the bottom layer: the "aa below" of that row), dm, the right-hand side
          for (int q = 0; q < kFL; q++) {
            vlds_st(B0, c0, q, aat[q]);
            vlds_st(B1, c0, q, dm[q]);
            vlds_st(B2, c0, q, d[q]);
          }
          for (int q = 0; q < kFL; q++)
            vlds_st_next_if(B0, c0, q, MOIST ? vdivq(t1g * gm2q[MOIST ? q : 0], dz[q]) * pemv[q + 1] : vdivq(vd(t1g * gm2), dz[q]) * pemv[q + 1], last[q]);
        }
      }
      // what the second half needs stays in the per-wavefront state (registers in the product build) across the barriers of EX
      // (dm and bb are re-formed from delp and grat with their own expressions; in EX the logarithms wait in the fourth LDS buffer,
      // which the w pass does not use: 50 registers less across the barriers, at two wavefronts per SIMD they were spilled)
      for (int q = 0; q < kFL; q++) {
        keep_pm2[s][q] = pm2[q]; keep_grat[s][q] = grat[q];
      }
      for (int q = 0; q <= kFL; q++) keep_pem[s][q] = pemv[q];
      if (!CG) {
        {
          for (int q = 0; q < kFL; q++) vlds_st(B3, c0, q, lnp[q]);
        }
      }
      if constexpr (SIM) {   // pp at the layer's top interface (0 at the model top) and at its bottom interface
        keep_ppt[s][0] = row_shr<1>(X[kFL - 1], 0.0);
        for (int q = 0; q < kFL; q++) keep_ppt[s][q + 1] = X[q];
      }
    }
    {
      FV3_TRACE(2);
      FV3_SYNC_LDS();
      FV3_TRACE(3);
      if (!(probe & 2)) w_columns(B0, B1, B2, pass_wave, tid);
      FV3_TRACE(4);
      FV3_SYNC_LDS();
      FV3_TRACE(5);
    }
    FV3_WAVE_FOR(wv) {
      const int s = FV3_WVI(wv), c0 = wv * 4;
      vd dm[kFL], pm2[kFL], grat[kFL], bb[kFL], lnp[kFL + 1], pemv[kFL + 1], w2[kFL];
      vb real[kFL], last[kFL];
      for (int q = 0; q < kFL; q++) {
        real[q] = vlevel_lt(q, km);
        last[q] = vlevel_eq(q, km - 1);
      }
      for (int q = 0; q < kFL; q++) {
        pm2[q] = keep_pm2[s][q]; grat[q] = keep_grat[s][q];
        dm[q] = dmr[s][q] * rgrav;
        bb[q] = vsel(real[q], vsel(last[q], vd(2.0), 2. * (1. + grat[q])), vd(1.0));
        w2[q] = vsel(real[q], vlds_ld(B2, c0, q), vd(0.0));
        if (cn.rff) w2[q] = w2[q] * vrow_ld(cn.rff, q, km);   // fast_tau_w_sec: w2(k) * rff(k) behind the back substitution (nh_utils.F90:1363-1371)
      }
      for (int q = 0; q <= kFL; q++) pemv[q] = keep_pem[s][q];
      if (!CG) {
        {
          for (int q = 0; q < kFL; q++) lnp[q] = vlds_ld(B3, c0, q);
          lnp[kFL] = row_shl<1>(lnp[0], 0.0);
        }
      }
      // ---- pe2 at the interfaces (:1373-1380): exclusive sum of dm2 (w2 - w1) / dt ----
      vd pe2[kFL + 2];
      {
        vd inc[kFL], tot(0.0);
        for (int q = 0; q < kFL; q++) {
          if constexpr (SIM)   // :1510-1514
            inc[q] = vsel(real[q], (dm[q] * (w2[q] - w1[s][q]) * rdt - beta * (keep_ppt[s][q + 1] - keep_ppt[s][q])) * ra, vd(0.0));
          else
            inc[q] = vsel(real[q], dm[q] * (w2[q] - w1[s][q]) * rdt, vd(0.0));
          tot = tot + inc[q];
        }
        vd run(0.0);
        {   // pe(k+1) = pe(k) + inc(k) from 0, in the reference's order
          vd in(0.0);
          for (int rnd = 0; rnd < nrounds; rnd++) {
            run = in;
            for (int q = 0; q < kFL; q++) {
              pe2[q] = run;
              run = run + inc[q];
            }
            in = row_shr<1>(run, 0.0);
          }
        }
        pe2[kFL] = row_shl<1>(pe2[0], 0.0);
        pe2[kFL + 1] = row_shl<1>(pe2[1], 0.0);
        // the interface below the lane's last layer when that is the model's bottom (lane 15 has no lane below)
        pe2[kFL] = vsel(vlevel_eq(kFL - 1, km - 1), run, pe2[kFL]);
      }
      // ---- p1 (upward recurrence, :1382-1392) and the new layer thickness ----
      vd dzn[kFL];
      {
        vd Aq[kFL], Bq[kFL];
        for (int q = 0; q < kFL; q++) {
          const vd Bl = (pe2[q] + 2. * pe2[q + 1]) * r3;
          const vd Bi = (pe2[q] + bb[q] * pe2[q + 1] + grat[q] * pe2[q + 2]) * r3;
          Aq[q] = vsel(real[q] && !last[q], -grat[q], vd(0.0));
          Bq[q] = vsel(real[q], vsel(last[q], Bl, Bi), vd(0.0));
        }
        vd p1v[kFL];
        {   // upwards from the bottom layer (its row takes nothing from below), in the reference's order
          vd in(0.0);
          for (int rnd = 0; rnd < nrounds; rnd++) {
            vd p1 = in;
            for (int q = kFL - 1; q >= 0; q--) {
              p1 = Bq[q] + Aq[q] * p1;
              p1v[q] = p1;
            }
            in = row_shl<1>(p1, 0.0);
          }
        }
        for (int q = kFL - 1; q >= 0; q--)
          dzn[q] = -dm[q] * rgas * ptv[s][q] * vexp(((MOIST ? cpv[s][MOIST ? q : 0] : vd(cp2)) - 1.) * vlog(vmax(cn.p_fac * pm2[q], p1v[q] + pm2[q])));
      }
      // ---- interface heights from the surface upwards (nh_core.F90:228-237 / nh_utils.F90:468-476) ----
      vd zn[kFL];
      {
        vd tot(0.0);
        for (int q = 0; q < kFL; q++) {
          dzn[q] = vsel(real[q], dzn[q], vd(0.0));
          tot = tot + (CG ? dzn[q] * cn.grav : dzn[q]);
        }
        const vd zsv = vcol_ld(zs, (long)o0, c0, ncol);
        {   // zh(k) = zh(k+1) - dz2(k) from the surface, in the reference's order (padded layers subtract 0)
          vd in = zsv;
          for (int rnd = 0; rnd < nrounds; rnd++) {
            vd run = in;
            for (int q = kFL - 1; q >= 0; q--) {
              run = run - (CG ? dzn[q] * cn.grav : dzn[q]);
              zn[q] = run;
            }
            in = row_shl1_v(run, zsv);
          }
        }
        // interface km (the surface) sits at (lane km / 8, q = km % 8): there every layer below is padding and run == zs
      }
      // ---- outputs through LDS, in the fields' own layouts ----
      for (int q = 0; q < kFL; q++) {
        vlds_st(B0, c0, q, zn[q]);
        if (CG) {
          vlds_st(B1, c0, q, vsel(vlevel_eq(q, 0), vd(cn.ptop), pe2[q] + pemv[q]));   // pef (:461-465)
        } else {
          vlds_st(B1, c0, q, w2[q]);
          vlds_st(B2, c0, q, dzn[q]);
        }
      }
      // keep what the second round needs
      if (!CG) {
        for (int q = 0; q < kFL; q++) {
          vd pef_ = pe2[q];
          if constexpr (SIM) pef_ = pe2[q] + beta * (keep_ppt[s][q] - pe2[q]);                              // :1531-1535
          dmr[s][q] = fp_out ? pef_ + pemv[q] : pef_;                                                      // ppe
          ptv[s][q] = vsel(vlevel_eq(q, 0), vd(dexp(cn.akap * dlog(cn.ptop))), vexp(cn.akap * lnp[q]));   // pk
          w1[s][q] = lnp[q];                                                                               // peln
          zv[s][q] = pemv[q];                                                                              // pe
        }
      }
    }
    FV3_TRACE(6);
    FV3_SYNC_LDS();
    FV3_TRACE(7);
    stage_out(B0, zl, ncol, km + 1, tid, [&](int col, int k) { return (ix_t)k * nA + o0 + col; });
    if (CG) {
      stage_out(B1, pef, ncol, km + 1, tid, [&](int col, int k) { return (ix_t)k * nA + o0 + col; });
      return;
    }
    const ix_t occ0 = (ix_t)g.iCC(i0, j);
    stage_out(B1, wq, ncol, km, tid, [&](int col, int k) { return (ix_t)k * nA + o0 + col; });
    stage_out(B2, delz, ncol, km, tid, [&](int col, int k) { return (ix_t)k * nCC + occ0 + col; });
    FV3_SYNC_LDS();
    FV3_WAVE_FOR(wv) {
      const int s = FV3_WVI(wv), c0 = wv * 4;
      for (int q = 0; q < kFL; q++) {
        vlds_st(B0, c0, q, dmr[s][q]);
        vlds_st(B1, c0, q, use_logp ? vsel(vlevel_eq(q, 0), ptv[s][q], w1[s][q]) : ptv[s][q]);   // pk3(1) = ptk either way (nh_core.F90:87)
        if (last_call) vlds_st(B2, c0, q, ptv[s][q]);
      }
    }
    FV3_SYNC_LDS();
    stage_out(B0, ppe, ncol, km + 1, tid, [&](int col, int k) { return (ix_t)k * nA + o0 + col; });
    stage_out(B1, pk3, ncol, km + 1, tid, [&](int col, int k) { return (ix_t)k * nA + o0 + col; });
    if (!last_call) return;
    stage_out(B2, pk, ncol, km + 1, tid, [&](int col, int k) { return (ix_t)k * nCC + occ0 + col; });
    FV3_SYNC_LDS();
    FV3_WAVE_FOR(wv) {
      const int s = FV3_WVI(wv), c0 = wv * 4;
      for (int q = 0; q < kFL; q++) {
        vlds_st(B0, c0, q, w1[s][q]);
        vlds_st(B1, c0, q, zv[s][q]);
      }
    }
    FV3_SYNC_LDS();
    stage_out(B0, peln, ncol, km + 1, tid, [&](int col, int k) {
      return (ix_t)(j - g.js) * g.nx * (km + 1) + (ix_t)k * g.nx + (i0 - g.is) + col; });
    stage_out(B1, pe, ncol, km + 1, tid, [&](int col, int k) {
      return (ix_t)(j - (g.js - 1)) * (g.nx + 2) * (km + 1) + (ix_t)k * (g.nx + 2) + (i0 - (g.is - 1)) + col; });
  }
#undef FV3_TRACE
};

}  // namespace fv3
