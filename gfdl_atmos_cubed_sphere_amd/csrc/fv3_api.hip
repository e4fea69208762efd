// fv3_api.hip -- the C ABI of include/fv3_mi355x.h: context, gridstruct upload, kernel launches.
// Compiled by hipcc --offload-arch=gfx950 into libfv3_mi355x.so (the product).  The same file is
// also compiled by g++ -DFV3_HOST_EMU under tests/hostemu (logic-test harness only).
#include "../../include/fv3_mi355x.h"

#include <algorithm>
#include <cstdarg>
#include <deque>
#include <unordered_map>
#include <new>
#include <type_traits>

#include "csw_kernel.h"
#include "csw_march.h"
#include "cubed_csw.h"
#include "cubed_tp.h"
#include "cubed_tpf.h"
#include "cube_topo.h"
#include "cubed_dsw.h"
#include "cubed_damp.h"
#include "cubed_a2b.h"
#include "dsw_kernels.h"
#include "dsw_march.h"
#include "dsw_fused.h"
#include "fv3_common.h"
#include "fv3_launch.h"
#include "nh_kernels.h"
#include "nh_fast.h"
#include "nh_alt.h"
#include "remap_kernels.h"
#include "remap_fast.h"
#include "tracer_kernels.h"
#include "tp2d_tile.h"

using namespace fv3;

// Tile shapes (cells per workgroup).  Tuned on MI355X; see DESIGN.md.
#ifndef FV3_CSW_TI
#define FV3_CSW_TI 32
#define FV3_CSW_TJ 8
#endif
#ifndef FV3_DSW_TI
#define FV3_DSW_TI 32
#define FV3_DSW_TJ 8
#endif

struct CubePlanDev;
struct fv3_ctx {
  fv3_domain dom;
  Grid g;           // device view (pointers into dev_metrics)
  stream_t stream;
  // side stream: the few levels that take the LDS-tile kernels in d_sw (sponge layers) run concurrently with
  // the marching kernels of the other levels (different levels = disjoint data)
  stream_t stream2;
  void *ev_fork, *ev_join;
  // lanes (dsw_cubed, csw_cubed): lane 1 = launches go to stream2 -- the frame / sponge-level passes of a cubed-sphere face beside the
  // marching kernels of its interior.  ev_mid: a second main -> side dependency inside one routine.  A face group keeps the side
  // stream and the events on its first member.
  int lane = 0;
  void *ev_mid = nullptr;
  bool side_ok;          // transport and momentum route the same levels to the tile kernels
  // the two ways of dividing the levels between the marching and the LDS-tile kernels of d_sw: [0] strict (no damping branch in the
  // marching kernels), [1] with the sponge levels of the reference defaults (nord_k = 0, nord_w = 0: dsw_fused.h run_bf) on the marching
  // side -- the active one (klist, n_plain, ...) is chosen per d_sw call (lev_activate)
  struct LevSel { int *klist, *klist_m; int n_plain, n_damp, n_plain_m, n_rest_m; bool side_ok; } lev_sel[2] = {};
  int use_side;          // FV3_MI355X_SIDE_STREAM=0 disables
  int sponge_march;      // FV3_MI355X_SPONGE_MARCH=0: the sponge levels stay on the LDS-tile kernels
  int round_simds;       // SIMDs of the device (CUs x 4): x wavefronts per SIMD of a kernel = the wavefronts resident at once
                         // (balance_segments); FV3_MI355X_ROUND_SIMDS overrides, 0 = no balancing
  double *dev_metrics;   // one allocation holding every metric array
  bool grid_ready;
  // per-level d_sw coefficients on the device
  int *lev_i;      // 4*npz
  double *lev_d;   // 5*npz
  bool lev_ready;
  // optional per-kernel timing with HIP events on the launch stream (fv3_profile / fv3_profile_report)
  // nonhydrostatic path: dp_ref + edge_profile coefficients, scratch slabs (A x (npz+1) each)
  double *dp0;        // device, npz
  double *edge_dev;   // device, 4*npz: gk, bet, gam, 1 / bet; then EdgeProfileLds::kTabDoubles: the rows' table (nh_fast.h edge_rows)
  EdgeCoef ec;
  bool dp0_ready;
  double *scratch[8];
  double *ray_d;         // pm(k), rf(k) of Rayleigh_Friction
  // host-address field registry (fv3_registry_*): host array -> its device mirror and which of the two copies is current
  struct RegEntry { const void *host; void *dev; size_t bytes; bool dev_current, host_current; };
  std::vector<RegEntry> *reg;
  int reg_lazy;
  long long reg_stat[4];   // h2d copies, h2d skipped, d2h copies, d2h deferred
  double *rff_d;         // rff(k) of fast_tau_w_sec (npz, 1.0 below k_rf) or null; rf(k), dp(k) of Ray_fast behind it (fv3_set_ray_fast)
  int rff_on, rayf_kmax, rayf_krf;
  double rayf_dm;
  bool moist_on;         // fv3_set_moist: moist thermodynamics of the remap
  bool remap_te_on;      // fv3_set_remap_te: total energy remapped in the place of T_v / theta_v
  const double *rte_hs;  // A
  double *rte_te;        // A x npz work array
  fv3_moist_params moist;
  double *moist_qcon, *moist_cappa;
  const double *q_con, *cappa;  // fv3_set_condensate: use_cond / moist_kappa arrays of the Riemann solvers (or null)
  double *remap_scr;     // coordinate + profile slabs of the vertical remap (fv3_lagrangian_to_eulerian)
  size_t remap_scr_n;
  double *lev_ext_d;  // damp(npz+1) for update_dz_d
  int *lev_ext_i;     // ndif(npz+1)
  double *trc_d;      // device, 2*npz: cmax, frac
  int *trc_i;         // device, npz: ksplt
  double *akbk;       // device, 2*(npz+1)
  int *kord_tr_dev;   // device, up to 64 tracers
  bool akbk_ready;
  // levels with del-2n damping of delp / w / pt go to the LDS-tile transport kernel, the others march
  int *klist;            // device, npz: [plain levels..., damped levels...]
  int n_plain, n_damp;
  // same split for the momentum part: marching needs nord_k == 1, no vorticity damping, d_con = 0
  int *klist_m;
  int n_plain_m, n_rest_m;
  int *klist_z;          // npz+1 interfaces of update_dz_d: [undamped..., damped...]
  int n_plain_z, n_damp_z;
  // peer exchange (fv3_comm_*, fv3_halo_start / _complete): RCCL communicator, its stream, events, message buffers
  void *comm;
  int comm_rank, comm_size;
  stream_t comm_stream;
  void *ev_packed, *ev_arrived;
  double *msg_send[8], *msg_recv[8];
  size_t msg_cap[8];
  int pend_n;
  // cube-edge exchange (fv3_cube_halo_start / _complete): per-kind pack / unpack plans of this face, message buffers, pending group
  struct CubePlanDev *cube_plan[5];
  int cube_face;
  double *cube_send, *cube_recv;
  size_t cube_cap_send, cube_cap_recv;
  int cube_pend_n;
  fv3_cube_field cube_pend[FV3_HALO_MAX_FIELDS];
  size_t cube_roff[FV3_HALO_MAX_FIELDS][6];
  fv3_halo_field pend_fields[FV3_HALO_MAX_FIELDS];
  int col_pool;      // workgroups of the pooled launches of the column solvers (0: one workgroup per 256 columns)
  int cubed_frame;   // cubed-sphere hybrid: width of the frame the pass kernels own (0: passes on the whole face)
  int cubed_reach;   // ... and how much wider the frame of the passes' intermediates is
  int cubed_frame_c; // the frame of c_sw (d2a2c_vect has its edge forms within 4 points of an edge)
  int lev_max_nord;      // max over the levels of nord_k
  int lev_max_nord_v, lev_max_nord_w, lev_max_nord_t;   // ... of nord_v / nord_w / nord_t over the levels where the damping is on
  bool lev_has_damp_v4, lev_has_damp_v5, lev_has_damp_t;  // damp_vt > 1e-4 (deln of delp) / > 1e-5 (del6 of vorticity); damp_t > 1e-4
  bool lev_has_dcon;     // some level has d_con_k > 1e-5
  bool lev_has_vt_damp, lev_has_w_damp, lev_has_w_damp_hi;  // damp_vt / damp_t; damp_w > 1e-5; the latter with nord_w > 0
  double *ke_scr;        // B kind, npz levels: KE + damping term at the corners
  double *mflux[2];      // mass-flux scratch of the marching transports: FX kind, FY kind (npz levels)
  double *heat_scr[2];   // heat_s / diss_e of a d_sw call whose caller passed NULL and whose levels are not all on the branch-free kernels
  // cubed sphere (grid_type < 3): edge weights / corner factors and the work arrays of the pass kernels (B x (npz+1) each)
  CubedGeom cg;
  double *cg_dev;
  double *cs_scr[36];
  int *ones_i;    // npz ones, device (ksplt of the inline_q sub-step)
  std::vector<double> host_area;   // prt_maxmin: area on the host, and g_sum's global_area
  double global_area = 0.;
  int march_tj;          // rows per wavefront segment of the marching kernels
  int march_tj_csw, march_tj_ke, march_tj_fused, march_tj_mom;
  int trc_nt;  // tracers per wavefront in the sub-cycle kernel (FV3_MI355X_TRACER_NT: 1..4, default 3)
  int remap_nt;  // tracers per thread in the remap (FV3_MI355X_REMAP_NT: 1..3, default 3)
  int riem_blocked;   // the same for the Riemann solvers' four slabs (FV3_MI355X_RIEM_SCR: 0 / 1, default 1)
  int pgrad_fused;    // nh_p_grad as ONE kernel where the domain has no face edges (NhPGradFused; FV3_MI355X_PGRAD_FUSED=0: a2b_ord4 + the gradient)
  int riem_lds;       // the dry SIM1 Riemann solvers with the levels across the lanes, BIT-IDENTICAL to the slab kernels (nh_fast.h
                      // RiemFast<CG, true>; FV3_MI355X_RIEM_LDS: 0 / 1, default 1)
  int remap_blocked;  // scratch slabs of the remap in per-wavefront blocks (FV3_MI355X_REMAP_SCR: 0 / 1, default 1)
  int remap_lds;      // the remap with the column in LDS (remap_fast.h; bit-identical to the slab kernels) where it is built for the
                      // configuration (FV3_MI355X_REMAP_LDS: 0 / 1, default 1)
  int tj_fixed;          // an FV3_MI355X_MARCH_TJ* variable is set: take the rows per segment as given
  int tj_env_fused;      // ... one of the d_sw kernels' (FV3_MI355X_MARCH_TJ_FUSED / _MOM)
  int csw_kpw;           // levels per wavefront in CswMarch (1 or 2; FV3_MI355X_CSW_KPW)
  int use_fused;         // 1: delp + w + pt in one marching kernel when the schemes allow (FV3_MI355X_FUSED=0: off)
  int use_march;         // 0: LDS-tile kernels only (FV3_MI355X_MARCH=0)
  bool prof_on;
  struct ProfRec { const char *label; void *e0, *e1; };
  std::vector<ProfRec> prof;
  struct fv3_group *grp = nullptr;   // the faces one rank holds, launched together (fv3_group_create)
  int grp_idx = 0;
};

static thread_local std::string g_err;

static int fail(const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return 1;
}
#define RT(call)                                                              \
  do {                                                                        \
    int e_ = (call);                                                          \
    if (e_) return fail("%s failed: %s (%s:%d)", #call, rt_errstr(e_), __FILE__, __LINE__); \
  } while (0)

extern "C" const char *fv3_last_error(void) { return g_err.c_str(); }

// the event pair of a profiled launch; a failure is reported through fv3_last_error and leaks nothing
static int prof_events(void **e0, void **e1) {
  int rc = rt_event_create(e0);
  if (rc) return fail("profiling event: %s", rt_errstr(rc));
  rc = rt_event_create(e1);
  if (rc) {
    rt_event_destroy(*e0);
    *e0 = nullptr;
    return fail("profiling event: %s", rt_errstr(rc));
  }
  return 0;
}

// ---- face groups: one launch per kernel for all faces a rank holds (include/fv3_mi355x.h fv3_group_create) -------------------------
// The host code issues every call once per face (tools/fv_mp_mod.F90's tiles of one PE; here six contexts on one MI355X).  A context
// that belongs to a group does not launch: it queues the launch -- functor by value, grid, label -- and when every member's queue
// holds the same kernel with the same grid at its head, ONE kernel runs them all (fv3_launch.h launch_group: the face is the slowest
// grid index).  Stream operations of a member (memset, device copies) queue as well and run in that member's order.  Whatever reads
// results or orders the stream against something else -- a download, a synchronisation, an event, a halo exchange -- flushes the queues
// first (what is left runs face by face).  All members launch on the first member's stream.
struct GrpEntry {
  const void *tag = nullptr;   // launch kind + functor type; nullptr: a stream operation, never merged
  int a = 0;                   // lanes (column kernels) / wavefronts (wave kernels)
  Dim3 grid{0, 0, 0};
  size_t lds = 0;
  const char *label = "";
  std::vector<double> fbuf;    // the functor's bytes
  int (*go)(struct fv3_group *, GrpEntry *const *, int, stream_t) = nullptr;
  int op = 0;                  // 1 memset, 2 device copy, 3 / 4 / 5 lane_exec (fork, mid, join)
  int lane = 0;                // 0: the group's stream, 1: its side stream (fv3_ctx::lane when the entry was queued)
  void *dst = nullptr;
  const void *src = nullptr;
  size_t bytes = 0;
  int value = 0;
};
struct fv3_group {
  int n = 0;
  fv3_ctx *m[kGrpMax];
  std::deque<GrpEntry> q[kGrpMax];
  long n_merged = 0, n_single = 0;
};
static std::vector<fv3_group *> g_groups;

template <class F, int KIND, int W>
static const void *grp_tag() {
  static const char t = 0;
  return &t;
}
template <class F, int KIND, int W>
static int grp_go(fv3_group *g, GrpEntry *const *e, int n, stream_t s) {
  (void)g;
  const F *fs[kGrpMax];
  for (int m = 0; m < n; m++) fs[m] = reinterpret_cast<const F *>(e[m]->fbuf.data());
  if constexpr (KIND == 2)
    return launch_group_cols<W>(e[0]->grid, e[0]->a, s, fs, n);
  else
    return launch_group<KIND>(e[0]->grid, e[0]->lds, e[0]->a, s, fs, n);
}
// the dependencies between the two lanes of a context (or of a group's first member): 3 fork -- the side stream waits for what the
// main stream holds now; 4 mid -- the same, a second time; 5 join -- the main stream waits for the side stream
enum { kLaneFork = 3, kLaneMid = 4, kLaneJoin = 5 };
static void lane_exec(fv3_ctx *o, int op) {
  if (op == kLaneJoin) {
    rt_event_record(o->ev_join, o->stream2);
    rt_stream_wait_event(o->stream, o->ev_join);
  } else {
    void *ev = op == kLaneFork ? o->ev_fork : o->ev_mid;
    rt_event_record(ev, o->stream);
    rt_stream_wait_event(o->stream2, ev);
  }
}
static int grp_run(fv3_group *g, fv3_ctx *owner, GrpEntry *const *e, int n) {   // n entries of one kernel (or one stream operation)
  const stream_t s = e[0]->lane ? g->m[0]->stream2 : g->m[0]->stream;
  if (!e[0]->tag) {
    if (e[0]->op >= kLaneFork) { lane_exec(g->m[0], e[0]->op); return 0; }
    if (e[0]->op == 1) return rt_memset(e[0]->dst, e[0]->value, e[0]->bytes, s);
    return rt_d2d(e[0]->dst, e[0]->src, e[0]->bytes, s);
  }
  void *e0 = nullptr, *e1 = nullptr;
  if (owner->prof_on) {
    if (prof_events(&e0, &e1)) return 1;
    rt_event_record(e0, s);
  }
  const int rc = e[0]->go(g, e, n, s);
  if (owner->prof_on) {
    rt_event_record(e1, s);
    owner->prof.push_back({e[0]->label, e0, e1});
  }
  if (n > 1) g->n_merged++; else g->n_single++;
  return rc;
}
// force: run what is queued even where the members are not at the same kernel (face by face, each member in its own order)
static int grp_pump(fv3_group *g, bool force) {
  for (;;) {
    int nonempty = 0;
    for (int m = 0; m < g->n; m++) nonempty += !g->q[m].empty();
    if (!nonempty) return 0;
    if (nonempty == g->n) {
      GrpEntry *h[kGrpMax];
      bool same = true;
      for (int m = 0; m < g->n; m++) {
        h[m] = &g->q[m].front();
        same = same && h[m]->tag && h[m]->tag == h[0]->tag && h[m]->a == h[0]->a && h[m]->lds == h[0]->lds && h[m]->lane == h[0]->lane &&
               h[m]->grid.x == h[0]->grid.x && h[m]->grid.y == h[0]->grid.y && h[m]->grid.z == h[0]->grid.z;
      }
      if (same) {
        const int rc = grp_run(g, g->m[0], h, g->n);
        for (int m = 0; m < g->n; m++) g->q[m].pop_front();
        if (rc) return rc;
        continue;
      }
    } else if (!force) {
      return 0;
    }
    for (int m = 0; m < g->n; m++) {   // heads that do not match (or a forced flush): one step of every member that has work
      if (g->q[m].empty()) continue;
      GrpEntry *h = &g->q[m].front();
      const int rc = grp_run(g, g->m[m], &h, 1);
      g->q[m].pop_front();
      if (rc) return rc;
    }
  }
}
static int grp_flush_all() {
  for (fv3_group *g : g_groups)
    if (int rc = grp_pump(g, true)) return rc;
  return 0;
}
template <class F, int KIND, int W = 0>
static int grp_defer(fv3_ctx *c, const char *label, Dim3 grid, size_t lds, int a, const F &f) {
  static_assert(std::is_trivially_copyable<F>::value, "kernel functors are plain data");
  fv3_group *g = c->grp;
  g->q[c->grp_idx].emplace_back();
  GrpEntry &e = g->q[c->grp_idx].back();
  e.tag = grp_tag<F, KIND, W>();
  e.a = a; e.grid = grid; e.lds = lds; e.label = label; e.lane = c->lane;
  e.fbuf.resize((sizeof(F) + 7) / 8);
  std::memcpy(e.fbuf.data(), (const void *)&f, sizeof(F));
  e.go = &grp_go<F, KIND, W>;
  return grp_pump(g, false);
}
static int grp_stream_op(fv3_ctx *c, int op, void *dst, const void *src, size_t bytes, int value) {
  if (c && c->grp) {
    fv3_group *g = c->grp;
    g->q[c->grp_idx].emplace_back();
    GrpEntry &e = g->q[c->grp_idx].back();
    e.op = op; e.dst = dst; e.src = src; e.bytes = bytes; e.value = value; e.lane = c->lane;
    return grp_pump(g, false);
  }
  const stream_t s = c ? (c->lane ? c->stream2 : c->stream) : nullptr;
  return op == 1 ? rt_memset(dst, value, bytes, s) : rt_d2d(dst, src, bytes, s);
}
// ---- lanes ------------------------------------------------------------------------------------------------------------------------
static stream_t lane_stream(const fv3_ctx *c) { return c->lane ? c->stream2 : c->stream; }
// the side stream and its events, on the context itself or on the first member of its group; created on first use
// When the two lanes pay (measured on C384 L127, tools/lanes_check.py): a face that launches on its own (one face per GPU) and is large
// enough for its marching kernels to last -- 4.5 -> 4.1 ms per pair; C96 L32: no gain; six faces in one launch (fv3_group): the pass
// launches are six times larger and fill the chip themselves, +-1 %, and at C96 the events cost 5 %.  FV3_MI355X_SIDE_STREAM: 0 never,
// 1 this policy, 2 always (experiments).
static bool lanes_pay(const fv3_ctx *c) {
  if (!c->use_side || c->prof_on) return false;
  if (c->use_side >= 2) return true;
  return !c->grp && (long)c->g.nx * c->g.ny * c->g.npz >= 2000000L;
}
static int lane_prepare(fv3_ctx *c) {
  fv3_ctx *o = c->grp ? c->grp->m[0] : c;
  // (a stream of its own for the lanes, of low priority: the marching kernels of the main stream are the critical path, the passes fill
  // what they leave; the periodic path's stream2 has the same role)
  if (!o->stream2) {
    static const int low = [] { const char *e = std::getenv("FV3_MI355X_SIDE_PRIO"); return e ? std::atoi(e) : 1; }();   // 0: normal priority
    if (int rc = low ? rt_stream_create_low(&o->stream2) : rt_stream_create(&o->stream2)) return rc;
  }
  if (!o->ev_fork) { if (int rc = rt_event_create(&o->ev_fork)) return rc; }
  if (!o->ev_join) { if (int rc = rt_event_create(&o->ev_join)) return rc; }
  if (!o->ev_mid) { if (int rc = rt_event_create(&o->ev_mid)) return rc; }
  return 0;
}
static int lane_op(fv3_ctx *c, int op) {   // in a group: queued, so that it keeps its place among the member's launches
  if (c->grp) {
    fv3_group *g = c->grp;
    g->q[c->grp_idx].emplace_back();
    g->q[c->grp_idx].back().op = op;
    return grp_pump(g, false);
  }
  lane_exec(c, op);
  return 0;
}
// stream calls that order the stream against the host or another stream: the queues of every group go first
static int rtf_h2d(void *d, const void *s, size_t n, stream_t st) { if (int rc = grp_flush_all()) return rc; return rt_h2d(d, s, n, st); }
static int rtf_d2h(void *d, const void *s, size_t n, stream_t st) { if (int rc = grp_flush_all()) return rc; return rt_d2h(d, s, n, st); }
static int rtf_sync(stream_t st) { if (int rc = grp_flush_all()) return rc; return rt_sync(st); }
// (int: a queued group launch that fails while it is flushed here must not be lost -- the exchange would pack and send stale data)
static int rtf_event_record(void *e, stream_t st) { if (int rc = grp_flush_all()) return rc; rt_event_record(e, st); return 0; }
static int rtf_stream_wait_event(stream_t st, void *e) { if (int rc = grp_flush_all()) return rc; rt_stream_wait_event(st, e); return 0; }

// launch + optional event pair around it
template <class F>
static int launch_p(fv3_ctx *c, const char *label, Dim3 grid, size_t lds_doubles, const F &f) {
  if (c->grp) return grp_defer<F, 0>(c, label, grid, lds_doubles, 0, f);
  void *e0 = nullptr, *e1 = nullptr;
  if (c->prof_on) {
    if (prof_events(&e0, &e1)) return 1;
    rt_event_record(e0, lane_stream(c));
  }
  int rc = launch(grid, lds_doubles, lane_stream(c), f);
  if (c->prof_on) {
    rt_event_record(e1, lane_stream(c));
    c->prof.push_back({label, e0, e1});
  }
  return rc;
}

template <class F>
static int launch_p2(fv3_ctx *c, const char *label, Dim3 grid, size_t lds_doubles, const F &f) {
  if (c->grp) return grp_defer<F, 1>(c, label, grid, lds_doubles, 0, f);
  void *e0 = nullptr, *e1 = nullptr;
  if (c->prof_on) {
    if (prof_events(&e0, &e1)) return 1;
    rt_event_record(e0, lane_stream(c));
  }
  int rc = launch_2w(grid, lds_doubles, lane_stream(c), f);
  if (c->prof_on) {
    rt_event_record(e1, lane_stream(c));
    c->prof.push_back({label, e0, e1});
  }
  return rc;
}

template <int W = 0, class F>
static int launch_c(fv3_ctx *c, const char *label, Dim3 grid, const F &f, int lanes = 0) {
  if (c->grp) return grp_defer<F, 2, W>(c, label, grid, 0, lanes, f);
  void *e0 = nullptr, *e1 = nullptr;
  if (c->prof_on) {
    if (prof_events(&e0, &e1)) return 1;
    rt_event_record(e0, lane_stream(c));
  }
  int rc = launch_cols<W>(grid, lane_stream(c), f, lanes);
  if (c->prof_on) {
    rt_event_record(e1, lane_stream(c));
    c->prof.push_back({label, e0, e1});
  }
  return rc;
}

template <class F>
static int launch_w(fv3_ctx *c, const char *label, int nwaves, const F &f) {
  if (c->grp) return grp_defer<F, 3>(c, label, Dim3{0, 0, 0}, 0, nwaves, f);
  void *e0 = nullptr, *e1 = nullptr;
  if (c->prof_on) {
    if (prof_events(&e0, &e1)) return 1;
    rt_event_record(e0, lane_stream(c));
  }
  int rc = launch_waves(nwaves, lane_stream(c), f);
  if (c->prof_on) {
    rt_event_record(e1, lane_stream(c));
    c->prof.push_back({label, e0, e1});
  }
  return rc;
}

extern "C" int fv3_profile(fv3_ctx *c, int enable) {
  if (!c) return fail("fv3_profile: null ctx");
  c->prof_on = enable != 0;
  return 0;
}

// the reference's timer (FMS mpp_clock through timing_on / timing_off, model/dyn_core.F90, fv_dynamics.F90, fv_tracer2d.F90) a
// kernel label of this library belongs to
static const char *reference_timer(const char *label) {
  auto starts = [&](const char *p) { return std::strncmp(label, p, std::strlen(p)) == 0; };
  if (starts("c_sw") || starts("cswc_")) return "C_SW";                                   // dyn_core.F90:437
  if (starts("d_sw") || starts("dswc_") || starts("inline_q") || starts("flux_accum")) return "D_SW";   // :659
  if (starts("update_dz_c")) return "UPDATE_DZ_C";                                          // :512
  if (starts("edge_profile") || starts("zh_") || starts("zhc_")) return "UPDATE_DZ";       // :909
  if (starts("riem_solver")) return "Riem_Solver";                                          // :529, :925
  if (starts("nh_p_grad") || starts("a2b") || starts("one_grad_p") || starts("divg2")) return "PG_D";   // :1015
  if (starts("halo_") || starts("cube_") || starts("gather")) return "COMM_TOTAL";
  if (starts("tracer_") || starts("trc_")) return "tracer_2d";                              // fv_dynamics.F90:521
  if (starts("fill2d")) return "Fill2D";                                                    // :543
  if (starts("remap_") || starts("energy_fixer") || starts("remap_finish")) return "Remapping";   // :571
  return "DYN_CORE";   // what dyn_core runs outside its inner timers (p_grad_c, geopk, pk3_halo, heating, omega ...)
}

static int profile_report_impl(fv3_ctx *c, char *out, size_t cap, bool timers) {
  if (!c || !out || cap == 0) return fail("fv3_profile_report: bad argument");
  RT(rtf_sync(c->stream));
  struct Acc { const char *label; int n; double ms; };
  std::vector<Acc> acc;
  for (auto &r : c->prof) {
    const double ms = rt_event_elapsed_ms(r.e0, r.e1);
    rt_event_destroy(r.e0);
    rt_event_destroy(r.e1);
    const char *label = timers ? reference_timer(r.label) : r.label;
    bool found = false;
    for (auto &a : acc)
      if (std::strcmp(a.label, label) == 0) { a.n++; a.ms += ms; found = true; break; }
    if (!found) acc.push_back({label, 1, ms});
  }
  c->prof.clear();
  std::string txt;
  char line[256];
  for (auto &a : acc) {
    snprintf(line, sizeof line, "%s %d %.6f\n", a.label, a.n, a.ms);
    txt += line;
  }
  if (txt.size() + 1 > cap) return fail("fv3_profile_report: buffer too small");
  std::memcpy(out, txt.c_str(), txt.size() + 1);
  return 0;
}
extern "C" int fv3_profile_report(fv3_ctx *c, char *out, size_t cap) { return profile_report_impl(c, out, cap, false); }
extern "C" int fv3_profile_report_timers(fv3_ctx *c, char *out, size_t cap) { return profile_report_impl(c, out, cap, true); }

extern "C" int fv3_create(const fv3_domain *dom, fv3_ctx **out) {
  if (!dom || !out) return fail("fv3_create: null argument");
  if (dom->ng != NG) return fail("fv3_create: ng must be %d", NG);
  if (dom->grid_type == 3 || dom->grid_type < 0)
    return fail("fv3_create: grid_type=%d not supported (4 = doubly periodic, 0..2 = cubed sphere)", dom->grid_type);
  if (dom->grid_type < 3 && (dom->is != 1 || dom->js != 1 || dom->ie != dom->npx - 1 || dom->je != dom->npy - 1 ||
                             dom->npx != dom->npy))
    return fail("fv3_create: a cubed-sphere context is one whole face (layout 1 x 1 per tile): is = js = 1, ie = je = npx - 1");
  if (dom->ie < dom->is || dom->je < dom->js || dom->npz < 1) return fail("fv3_create: empty domain");
  fv3_ctx *c = new (std::nothrow) fv3_ctx();
  if (!c) return fail("fv3_create: out of host memory");
  c->dom = *dom;
  Grid &g = c->g;
  std::memset(&g, 0, sizeof g);
  g.is = dom->is; g.ie = dom->ie; g.js = dom->js; g.je = dom->je;
  g.isd = dom->is - NG; g.ied = dom->ie + NG; g.jsd = dom->js - NG; g.jed = dom->je + NG;
  g.npx = dom->npx; g.npy = dom->npy; g.npz = dom->npz;
  g.nid = g.ied - g.isd + 1; g.njd = g.jed - g.jsd + 1; g.nx = g.ie - g.is + 1; g.ny = g.je - g.js + 1;
  g.grid_type = dom->grid_type;
  g.do_diss_est = dom->do_diss_est; g.prevent_diss_cooling = dom->prevent_diss_cooling;
  g.stretched_grid = dom->stretched_grid;
  g.lim_fac = dom->lim_fac;
  c->stream = nullptr;
  c->stream2 = nullptr; c->ev_fork = c->ev_join = nullptr; c->side_ok = false;
  c->dev_metrics = nullptr;
  c->grid_ready = false;
  c->lev_i = nullptr; c->lev_d = nullptr; c->lev_ready = false;
  c->prof_on = false;
  c->klist = nullptr; c->n_plain = c->n_damp = 0;
  c->klist_m = nullptr; c->n_plain_m = c->n_rest_m = 0; c->ke_scr = nullptr;
  c->klist_z = nullptr; c->n_plain_z = c->n_damp_z = 0;
  c->mflux[0] = c->mflux[1] = nullptr;
  c->heat_scr[0] = c->heat_scr[1] = nullptr;
  std::memset(&c->cg, 0, sizeof c->cg);
  c->cg_dev = nullptr;
  for (auto &p : c->cs_scr) p = nullptr;
  c->comm = nullptr; c->comm_rank = 0; c->comm_size = 1; c->comm_stream = nullptr; c->ev_packed = c->ev_arrived = nullptr;
  for (int d = 0; d < 8; d++) { c->msg_send[d] = c->msg_recv[d] = nullptr; c->msg_cap[d] = 0; }
  c->pend_n = 0;
  for (int n = 0; n < 5; n++) c->cube_plan[n] = nullptr;
  c->cube_face = -1; c->cube_send = c->cube_recv = nullptr; c->cube_cap_send = c->cube_cap_recv = 0; c->cube_pend_n = 0;
  {  // tuning / fallback knobs
    const char *e = std::getenv("FV3_MI355X_MARCH");
    c->use_march = e ? std::atoi(e) : 1;
    e = std::getenv("FV3_MI355X_SIDE_STREAM");
    c->use_side = e ? std::atoi(e) : 1;
    e = std::getenv("FV3_MI355X_SPONGE_MARCH");
    c->sponge_march = e ? std::atoi(e) : 1;
    c->round_simds = 0;
#ifndef FV3_HOST_EMU
    {
      int dev = 0, ncu = 0;
      if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess)
        c->round_simds = 4 * ncu;
    }
#endif
    e = std::getenv("FV3_MI355X_ROUND_SIMDS");
    if (e) c->round_simds = std::atoi(e);
    c->tj_fixed = 0;
    for (const char *v : {"FV3_MI355X_MARCH_TJ", "FV3_MI355X_MARCH_TJ_FUSED", "FV3_MI355X_MARCH_TJ_MOM",
                          "FV3_MI355X_MARCH_TJ_KE", "FV3_MI355X_MARCH_TJ_CSW"})
      if (std::getenv(v)) c->tj_fixed = 1;
    c->tj_env_fused = (std::getenv("FV3_MI355X_MARCH_TJ_FUSED") || std::getenv("FV3_MI355X_MARCH_TJ_MOM")) ? 1 : 0;
    e = std::getenv("FV3_MI355X_MARCH_TJ");
    c->march_tj = e ? std::atoi(e) : 48;
    if (c->march_tj < 1) c->march_tj = 48;
    e = std::getenv("FV3_MI355X_CSW_KPW");
    c->csw_kpw = e ? std::atoi(e) : 0;   // 0 = by geometry mode (fv3_c_sw)
    if (c->csw_kpw < 0 || c->csw_kpw > 2) c->csw_kpw = 0;   // (three and four levels per wavefront were measured in round 1 and never won)
    e = std::getenv("FV3_MI355X_FUSED");
    c->use_fused = e ? std::atoi(e) : 1;
    e = std::getenv("FV3_MI355X_MARCH_TJ_FUSED");
    // 48 rows (eight segments of a 384-row tile): measured on one set of arrays with all 127 levels on the marching kernels -- 0.773 ms
    // against 0.794 for seven segments of 55 rows trimmed to whole rounds and 0.786 / 0.780 for ten / nine (tools/pair_ab2.py)
    c->march_tj_fused = e ? std::atoi(e) : 48;
    if (c->march_tj_fused < 1) c->march_tj_fused = 48;
    e = std::getenv("FV3_MI355X_MARCH_TJ_MOM");
    c->march_tj_mom = e ? std::atoi(e) : c->march_tj_fused;
    if (c->march_tj_mom < 1) c->march_tj_mom = c->march_tj_fused;
    e = std::getenv("FV3_MI355X_TRACER_NT");
    c->trc_nt = e ? std::atoi(e) : 3;
    if (c->trc_nt < 1 || c->trc_nt > 4) c->trc_nt = 3;
    e = std::getenv("FV3_MI355X_REMAP_NT");
    c->remap_nt = e ? std::atoi(e) : 3;
    if (c->remap_nt < 1 || c->remap_nt > RemapFields::kGroupMax) c->remap_nt = 3;
    e = std::getenv("FV3_MI355X_RIEM_SCR");
    c->riem_blocked = e ? (std::atoi(e) != 0) : 1;
    e = std::getenv("FV3_MI355X_RIEM_LDS");
    c->riem_lds = e ? (std::atoi(e) != 0) : 1;
    const char *ef = std::getenv("FV3_MI355X_PGRAD_FUSED");
    c->pgrad_fused = ef ? (std::atoi(ef) != 0) : 1;
    e = std::getenv("FV3_MI355X_REMAP_SCR");
    c->remap_blocked = e ? (std::atoi(e) != 0) : 1;
    e = std::getenv("FV3_MI355X_REMAP_LDS");
    c->remap_lds = e ? std::atoi(e) : 1;   // 0: slab kernels, 1: the LDS kernels where they pay (fv3_lagrangian_to_eulerian), 2: wherever built
    // the levels-across-the-lanes kernels index the fields with 32 bits (nh_fast.h ix_t): a tile whose fields reach 2^32 bytes takes the
    // slab kernels (1030 x 1030 x 128 is 1.1e9 bytes)
    if ((size_t)g.nB() * (size_t)(g.npz + 2) >= ((size_t)1 << 29)) {
      c->riem_lds = 0;
      c->remap_lds = 0;
    }
    e = std::getenv("FV3_MI355X_MARCH_TJ_KE");
    c->march_tj_ke = e ? std::atoi(e) : 48;
    if (c->march_tj_ke < 1) c->march_tj_ke = 48;
    e = std::getenv("FV3_MI355X_COL_POOL");
    c->col_pool = e ? std::atoi(e) : 0;
    if (c->col_pool < 0) c->col_pool = 0;
    e = std::getenv("FV3_MI355X_CUBED_FRAME");
    c->cubed_frame = e ? std::atoi(e) : 4;
    if (c->cubed_frame < 0) c->cubed_frame = 0;
    e = std::getenv("FV3_MI355X_CUBED_FRAME_C");
    c->cubed_frame_c = e ? std::atoi(e) : (c->cubed_frame ? 7 : 0);
    if (c->cubed_frame_c < 0) c->cubed_frame_c = 0;
    e = std::getenv("FV3_MI355X_CUBED_REACH");
    c->cubed_reach = e ? std::atoi(e) : 5;
    if (c->cubed_reach < 1) c->cubed_reach = 5;
    e = std::getenv("FV3_MI355X_MARCH_TJ_CSW");
    c->march_tj_csw = e ? std::atoi(e) : 0;   // 0 = by geometry mode (fv3_c_sw)
    if (c->march_tj_csw < 0) c->march_tj_csw = 0;
  }
  c->dp0 = nullptr; c->edge_dev = nullptr; c->dp0_ready = false;
  c->akbk = nullptr; c->kord_tr_dev = nullptr; c->akbk_ready = false;
  c->remap_scr = nullptr; c->remap_scr_n = 0; c->ray_d = nullptr;
  c->reg = nullptr; c->reg_lazy = 0; c->reg_stat[0] = c->reg_stat[1] = c->reg_stat[2] = c->reg_stat[3] = 0;
  c->rff_d = nullptr; c->rff_on = 0; c->rayf_kmax = -1; c->rayf_krf = 0; c->rayf_dm = 1.;
  c->q_con = nullptr; c->cappa = nullptr;
  c->moist_on = false; c->moist_qcon = nullptr; c->moist_cappa = nullptr;
  c->remap_te_on = false; c->rte_hs = nullptr; c->rte_te = nullptr;
  c->trc_d = nullptr; c->trc_i = nullptr; c->ones_i = nullptr;
  for (auto &s : c->scratch) s = nullptr;
  c->lev_ext_d = nullptr; c->lev_ext_i = nullptr;
  *out = c;
  return 0;
}

extern "C" int fv3_comm_destroy(fv3_ctx *c);
static void cube_plans_free(fv3_ctx *c);
extern "C" int fv3_destroy(fv3_ctx *c) {
  if (!c) return 0;
  if (c->grp) {   // a member that goes away: its group runs what is queued and forgets it (the group then launches face by face)
    fv3_group *g = c->grp;
    (void)grp_pump(g, true);
    for (int m = 0; m < g->n; m++)
      if (g->m[m] == c) g->m[m] = nullptr;
    for (int m = 0; m < g->n; m++)
      if (g->m[m]) g->m[m]->grp = nullptr;      // no longer a whole group: its remaining members launch on their own
    g->n = 0;
    c->grp = nullptr;
  }
  cube_plans_free(c);
  fv3_comm_destroy(c);
  if (c->dev_metrics) rt_free(c->dev_metrics);
  if (c->lev_i) rt_free(c->lev_i);
  if (c->lev_d) rt_free(c->lev_d);
  if (c->dp0) rt_free(c->dp0);
  if (c->akbk) rt_free(c->akbk);
  if (c->remap_scr) rt_free(c->remap_scr);
  if (c->ray_d) rt_free(c->ray_d);
  if (c->rff_d) rt_free(c->rff_d);
  delete c->reg;
  if (c->trc_d) rt_free(c->trc_d);
  if (c->trc_i) rt_free(c->trc_i);
  if (c->ones_i) rt_free(c->ones_i);
  if (c->kord_tr_dev) rt_free(c->kord_tr_dev);
  if (c->edge_dev) rt_free(c->edge_dev);
  if (c->lev_ext_d) rt_free(c->lev_ext_d);
  if (c->lev_ext_i) rt_free(c->lev_ext_i);
  for (auto &s : c->scratch) if (s) rt_free(s);
  for (auto &s : c->mflux) if (s) rt_free(s);
  for (auto &s : c->cs_scr) if (s) rt_free(s);
  if (c->cg_dev) rt_free(c->cg_dev);
  if (c->stream2) rt_stream_destroy(c->stream2);
  if (c->ev_fork) rt_event_destroy(c->ev_fork);
  if (c->ev_join) rt_event_destroy(c->ev_join);
  if (c->ev_mid) rt_event_destroy(c->ev_mid);
  for (auto &ls : c->lev_sel) {
    if (ls.klist) rt_free(ls.klist);
    if (ls.klist_m) rt_free(ls.klist_m);
  }
  if (c->klist_z) rt_free(c->klist_z);
  if (c->ke_scr) rt_free(c->ke_scr);
  for (double *h : c->heat_scr) if (h) rt_free(h);
  delete c;
  return 0;
}

extern "C" int fv3_set_stream(fv3_ctx *c, void *stream) {
  if (!c) return fail("fv3_set_stream: null ctx");
  if (c->grp) {   // the faces of a group launch together: one stream for all of them
    RT(grp_pump(c->grp, true));
    for (int m = 0; m < c->grp->n; m++) c->grp->m[m]->stream = (stream_t)stream;
  }
  c->stream = (stream_t)stream;
  return 0;
}

// ---- fv3_group_*: the faces one rank holds, launched together --------------------------------------------------------------------
extern "C" int fv3_group_create(fv3_ctx *const *members, int n, fv3_group **out) {
  if (!members || !out || n < 1 || n > kGrpMax) return fail("fv3_group_create: 1 .. %d members", kGrpMax);
  for (int m = 0; m < n; m++) {
    if (!members[m]) return fail("fv3_group_create: null member");
    if (members[m]->grp) return fail("fv3_group_create: member %d belongs to a group already", m);
    for (int o = 0; o < m; o++)
      if (members[o] == members[m]) return fail("fv3_group_create: member %d twice", m);
  }
  fv3_group *g = new (std::nothrow) fv3_group();
  if (!g) return fail("fv3_group_create: out of host memory");
  g->n = n;
  for (int m = 0; m < n; m++) {
    g->m[m] = members[m];
    members[m]->grp = g;
    members[m]->grp_idx = m;
    members[m]->stream = members[0]->stream;
  }
  g_groups.push_back(g);
  *out = g;
  return 0;
}
extern "C" int fv3_group_flush(fv3_group *g) {
  if (!g) return fail("fv3_group_flush: null group");
  RT(grp_pump(g, true));
  return 0;
}
// launches since the last call: kernels that ran all members at once, kernels (and members' steps) that ran alone
extern "C" int fv3_group_stats(fv3_group *g, long *merged, long *single) {
  if (!g || !merged || !single) return fail("fv3_group_stats: null argument");
  *merged = g->n_merged; *single = g->n_single;
  g->n_merged = g->n_single = 0;
  return 0;
}
extern "C" int fv3_group_destroy(fv3_group *g) {
  if (!g) return 0;
  const int rc = grp_pump(g, true);
  for (int m = 0; m < g->n; m++)
    if (g->m[m]) g->m[m]->grp = nullptr;
  g_groups.erase(std::remove(g_groups.begin(), g_groups.end(), g), g_groups.end());
  delete g;
  return rc ? fail("fv3_group_destroy: a queued launch failed") : 0;
}

// Field arrays come back from hipMalloc aligned to 2 MB, i.e. every field starts at the same phase of the HBM channel interleave; a
// stencil kernel that streams six or ten fields at the same (i, j, k) then sends all its streams to the same channels at the same
// time.  FV3_MI355X_MALLOC_SKEW=S (bytes, a multiple of 256; default below) starts the n-th array n * S bytes (mod 64 KB) into its
// allocation, so the streams of a kernel sit at different phases.  0: as hipMalloc returns them.
static std::unordered_map<void *, void *> g_skewed;   // user pointer -> allocation
static size_t malloc_skew() {
  static const size_t v = [] {
    const char *e = std::getenv("FV3_MI355X_MALLOC_SKEW");
    long n = e ? std::atol(e) : 0;
    if (n < 0) n = 0;
    return (size_t)(n / 256 * 256);
  }();
  return v;
}
extern "C" int fv3_malloc(void **dptr, size_t bytes) {
#ifndef FV3_HOST_EMU
  const size_t sk = malloc_skew();
  if (sk && bytes >= (1u << 20)) {
    static size_t counter = 0;
    const size_t off = (counter++ * sk) % 65536;
    void *base = nullptr;
    RT(rt_malloc(&base, bytes + 65536));
    *dptr = static_cast<char *>(base) + off;
    g_skewed[*dptr] = base;
    return 0;
  }
#endif
  RT(rt_malloc(dptr, bytes));
  return 0;
}
extern "C" int fv3_free(void *dptr) {
  RT(grp_flush_all());   // a queued launch of a face group may still use the buffer
  auto it = g_skewed.find(dptr);
  if (it != g_skewed.end()) {
    void *base = it->second;
    g_skewed.erase(it);
    RT(rt_free(base));
    return 0;
  }
  RT(rt_free(dptr));
  return 0;
}
extern "C" int fv3_memcpy_h2d(fv3_ctx *c, void *dst, const void *src, size_t bytes) {
  RT(rtf_h2d(dst, src, bytes, c ? c->stream : nullptr));
  return 0;
}
extern "C" int fv3_memcpy_d2h(fv3_ctx *c, void *dst, const void *src, size_t bytes) {
  RT(rtf_d2h(dst, src, bytes, c ? c->stream : nullptr));
  return 0;
}
// ---- host-address field registry: the reference's entry points take HOST arrays the caller owns; a wrapper with the same
// argument list copies them in and out on every call unless it knows which copy is current.  An entry is keyed by the host address.
// Eager (the default): every put / get copies -- the caller may have touched anything.  Lazy (fv3_registry_mode(ctx, 1)): the caller
// declares what it wrote (fv3_registry_host_touched) and asks for what it reads (fv3_registry_fetch); everything else stays on the device.
static fv3_ctx::RegEntry *reg_find(fv3_ctx *c, const void *host, const void *dev) {
  if (!c->reg) c->reg = new std::vector<fv3_ctx::RegEntry>();
  for (auto &e : *c->reg)
    if ((host && e.host == host) || (!host && dev && e.dev == dev)) return &e;
  return nullptr;
}
extern "C" int fv3_registry_mode(fv3_ctx *c, int lazy) {
  if (!c) return fail("fv3_registry_mode: null context");
  c->reg_lazy = lazy ? 1 : 0;
  return 0;
}
extern "C" int fv3_registry_put(fv3_ctx *c, void *dev, const void *host, size_t bytes) {
  if (!c || !dev || !host) return fail("fv3_registry_put: null argument");
  fv3_ctx::RegEntry *e = reg_find(c, host, nullptr);
  if (e && (e->dev != dev || e->bytes != bytes)) {   // the host address is bound to another mirror now (a freed and reused array)
    e->dev = dev; e->bytes = bytes; e->dev_current = false; e->host_current = true;
  }
  if (!e) {
    c->reg->push_back(fv3_ctx::RegEntry{host, dev, bytes, false, true});
    e = &c->reg->back();
  }
  if (c->reg_lazy && e->dev_current) { c->reg_stat[1]++; return 0; }
  RT(rtf_h2d(dev, host, bytes, c->stream));
  e->dev_current = true;
  c->reg_stat[0]++;
  return 0;
}
extern "C" int fv3_registry_get(fv3_ctx *c, void *host, const void *dev, size_t bytes) {
  if (!c || !dev || !host) return fail("fv3_registry_get: null argument");
  fv3_ctx::RegEntry *e = reg_find(c, host, nullptr);
  if (!e) {
    c->reg->push_back(fv3_ctx::RegEntry{host, const_cast<void *>(dev), bytes, true, false});
    e = &c->reg->back();
  } else if (e->dev != dev || e->bytes != bytes) {
    e->dev = const_cast<void *>(dev); e->bytes = bytes; e->dev_current = true;
  }
  e->host_current = false;            // a kernel wrote the mirror since the host copy was current
  e->dev_current = true;
  if (c->reg_lazy) { c->reg_stat[3]++; return 0; }   // deferred: fv3_registry_fetch brings it when the caller reads it
  RT(rtf_d2h(host, dev, bytes, c->stream));
  e->host_current = true;
  c->reg_stat[2]++;
  return 0;
}
extern "C" int fv3_registry_host_touched(fv3_ctx *c, const void *host) {
  if (!c) return fail("fv3_registry_host_touched: null context");
  if (!c->reg) return 0;
  for (auto &e : *c->reg)
    if (!host || e.host == host) { e.dev_current = false; e.host_current = true; }
  return 0;
}
extern "C" int fv3_registry_fetch(fv3_ctx *c, void *host) {
  if (!c) return fail("fv3_registry_fetch: null context");
  if (!c->reg) return 0;
  bool any = false;
  for (auto &e : *c->reg)
    if ((!host || e.host == host) && !e.host_current) {
      RT(rtf_d2h(const_cast<void *>(e.host), e.dev, e.bytes, c->stream));
      e.host_current = true;
      c->reg_stat[2]++;
      any = true;
    }
  if (any) RT(rtf_sync(c->stream));
  return 0;
}
// an entry is keyed by the HOST ADDRESS and knows nothing of the array's lifetime: a host array that is freed and allocated again at the
// same address with the same mirror and size would still read as "device copy current" in lazy mode.  The caller that frees or rebinds
// an array says so (host = NULL: every entry); what the host copy lacks is fetched first unless `discard`.
extern "C" int fv3_registry_forget(fv3_ctx *c, void *host, int discard) {
  if (!c) return fail("fv3_registry_forget: null context");
  if (!c->reg) return 0;
  if (!discard) RT(fv3_registry_fetch(c, host));
  auto &v = *c->reg;
  for (size_t n = v.size(); n-- > 0;)
    if (!host || v[n].host == host) v.erase(v.begin() + (long)n);
  return 0;
}
extern "C" int fv3_registry_stats(fv3_ctx *c, long long *out4) {
  if (!c || !out4) return fail("fv3_registry_stats: null argument");
  for (int i = 0; i < 4; i++) out4[i] = c->reg_stat[i];
  return 0;
}
extern "C" int fv3_memcpy_d2d(fv3_ctx *c, void *dst, const void *src, size_t bytes) {
  RT(grp_stream_op(c, 2, dst, src, bytes, 0));
  return 0;
}
extern "C" int fv3_memset(fv3_ctx *c, void *dst, int value, size_t bytes) {
  RT(grp_stream_op(c, 1, dst, nullptr, bytes, value));
  return 0;
}
extern "C" int fv3_sync(fv3_ctx *c) {
  RT(rtf_sync(c ? c->stream : nullptr));
  return 0;
}

extern "C" int fv3_grid_upload(fv3_ctx *c, const fv3_grid_host *h) {
  if (!c || !h) return fail("fv3_grid_upload: null argument");
  c->host_area.clear();   // fv3_prt_maxmin's copies of the area and of g_sum's global area belong to the grid that goes away
  c->global_area = 0.;
  Grid &g = c->g;
  const size_t nA = g.nA(), nU = g.nU(), nV = g.nV(), nB = g.nB();
  struct Item { const double *src; const double **dst; size_t n; };
  Item items[] = {
      {h->area, &g.area, nA}, {h->rarea, &g.rarea, nA}, {h->dxa, &g.dxa, nA}, {h->dya, &g.dya, nA},
      {h->rdxa, &g.rdxa, nA}, {h->rdya, &g.rdya, nA}, {h->cosa_s, &g.cosa_s, nA}, {h->rsin2, &g.rsin2, nA},
      {h->f0, &g.f0, nA},
      {h->dx, &g.dx, nU}, {h->rdx, &g.rdx, nU}, {h->dyc, &g.dyc, nU}, {h->rdyc, &g.rdyc, nU},
      {h->cosa_v, &g.cosa_v, nU}, {h->sina_v, &g.sina_v, nU}, {h->rsin_v, &g.rsin_v, nU},
      {h->divg_u, &g.divg_u, nU}, {h->del6_u, &g.del6_u, nU},
      {h->dy, &g.dy, nV}, {h->rdy, &g.rdy, nV}, {h->dxc, &g.dxc, nV}, {h->rdxc, &g.rdxc, nV},
      {h->cosa_u, &g.cosa_u, nV}, {h->sina_u, &g.sina_u, nV}, {h->rsin_u, &g.rsin_u, nV},
      {h->divg_v, &g.divg_v, nV}, {h->del6_v, &g.del6_v, nV},
      {h->rarea_c, &g.rarea_c, nB}, {h->fC, &g.fC, nB}, {h->cosa, &g.cosa, nB}, {h->sina, &g.sina, nB},
      {h->sin_sg, &g.sin_sg, 9 * nA}, {h->cos_sg, &g.cos_sg, 9 * nA},
  };
  size_t total = 0;
  for (const Item &it : items) {
    if (!it.src) return fail("fv3_grid_upload: a metric pointer is null");
    total += (it.n + 7) & ~(size_t)7;
  }
  if (!c->dev_metrics) RT(rt_malloc((void **)&c->dev_metrics, total * sizeof(double)));
  size_t off = 0;
  for (const Item &it : items) {
    RT(rtf_h2d(c->dev_metrics + off, it.src, it.n * sizeof(double), c->stream));
    *it.dst = c->dev_metrics + off;
    off += (it.n + 7) & ~(size_t)7;
  }
  g.da_min = h->da_min;
  g.da_min_c = h->da_min_c;
  {  // geometry mode from the arrays themselves (Grid::geom); FV3_MI355X_GEOM caps it (0 = always the general kernels)
    auto all_eq = [](const double *p, size_t n, double v) {
      for (size_t i = 0; i < n; i++)
        if (p[i] != v) return false;
      return true;
    };
    bool ortho = all_eq(h->cosa_s, nA, 0.) && all_eq(h->rsin2, nA, 1.) && all_eq(h->cosa_u, nV, 0.) &&
                 all_eq(h->sina_u, nV, 1.) && all_eq(h->rsin_u, nV, 1.) && all_eq(h->cosa_v, nU, 0.) &&
                 all_eq(h->sina_v, nU, 1.) && all_eq(h->rsin_v, nU, 1.) && all_eq(h->sin_sg, 4 * nA, 1.);
    struct Uni { const double *src; size_t n; double *dst; };
    const Uni uni[] = {
        {h->area, nA, &g.c_area}, {h->rarea, nA, &g.c_rarea}, {h->dxa, nA, &g.c_dxa}, {h->dya, nA, &g.c_dya},
        {h->rdxa, nA, &g.c_rdxa}, {h->rdya, nA, &g.c_rdya}, {h->dx, nU, &g.c_dx}, {h->rdx, nU, &g.c_rdx},
        {h->dyc, nU, &g.c_dyc}, {h->rdyc, nU, &g.c_rdyc}, {h->dy, nV, &g.c_dy}, {h->rdy, nV, &g.c_rdy},
        {h->dxc, nV, &g.c_dxc}, {h->rdxc, nV, &g.c_rdxc}, {h->divg_u, nU, &g.c_divg_u}, {h->divg_v, nV, &g.c_divg_v},
        {h->del6_u, nU, &g.c_del6_u}, {h->del6_v, nV, &g.c_del6_v}, {h->rarea_c, nB, &g.c_rarea_c},
    };
    bool uniform = ortho;
    for (const Uni &u : uni) {
      *u.dst = u.src[0];
      uniform = uniform && all_eq(u.src, u.n, u.src[0]);
    }
    g.geom = uniform ? 2 : (ortho ? 1 : 0);
    if (const char *e = std::getenv("FV3_MI355X_GEOM")) g.geom = std::min(g.geom, std::max(0, std::atoi(e)));
  }
  RT(rtf_sync(c->stream));  // host buffers may go away after the call returns
  c->grid_ready = true;
  return 0;
}

extern "C" int fv3_grid_geom(const fv3_ctx *c) { return (c && c->grid_ready) ? c->g.geom : -1; }

static void lev_activate(fv3_ctx *c, int x) {
  const fv3_ctx::LevSel &ls = c->lev_sel[x];
  c->klist = ls.klist; c->n_plain = ls.n_plain; c->n_damp = ls.n_damp;
  c->klist_m = ls.klist_m; c->n_plain_m = ls.n_plain_m; c->n_rest_m = ls.n_rest_m;
  c->side_ok = ls.side_ok;
}

extern "C" int fv3_dsw_levels_upload(fv3_ctx *c, const fv3_dsw_levels *lv) {
  if (!c || !lv) return fail("fv3_dsw_levels_upload: null argument");
  const int npz = c->g.npz;
  if (!c->lev_i) RT(rt_malloc((void **)&c->lev_i, sizeof(int) * 4 * npz));
  if (!c->lev_d) RT(rt_malloc((void **)&c->lev_d, sizeof(double) * 5 * npz));
  const int *iv[4] = {lv->nord_k, lv->nord_v, lv->nord_w, lv->nord_t};
  const double *dv[5] = {lv->d2_divg, lv->damp_vt, lv->damp_w, lv->damp_t, lv->d_con_k};
  for (int n = 0; n < 4; n++) {
    for (int k = 0; k < npz; k++) {
      // divergence damping supports nord <= 3 (halo 3), deln/del6 damping nord <= 2 (sw_core.F90:1610)
      if (iv[n][k] < 0 || iv[n][k] > (n == 0 ? 3 : 2)) return fail("fv3_dsw_levels_upload: nord out of range at k=%d", k);
    }
    RT(rtf_h2d(c->lev_i + n * npz, iv[n], sizeof(int) * npz, c->stream));
  }
  for (int n = 0; n < 5; n++) RT(rtf_h2d(c->lev_d + n * npz, dv[n], sizeof(double) * npz, c->stream));
  {  // ndif(km+1), damp(km+1) of update_dz_d: entry km+1 repeats entry km (nh_utils.F90:240-241)
    std::vector<int> ni(npz + 1);
    std::vector<double> nd(npz + 1);
    for (int k = 0; k < npz; k++) { ni[k] = lv->nord_v[k]; nd[k] = lv->damp_vt[k]; }
    ni[npz] = ni[npz - 1];
    nd[npz] = nd[npz - 1];
    if (!c->lev_ext_i) RT(rt_malloc((void **)&c->lev_ext_i, sizeof(int) * (npz + 1)));
    if (!c->lev_ext_d) RT(rt_malloc((void **)&c->lev_ext_d, sizeof(double) * (npz + 1)));
    std::vector<int> pz, dz;
    for (int k = 0; k <= npz; k++) ((nd[k] > 1.E-5) ? dz : pz).push_back(k);
    c->n_plain_z = (int)pz.size();
    c->n_damp_z = (int)dz.size();
    pz.insert(pz.end(), dz.begin(), dz.end());
    if (!c->klist_z) RT(rt_malloc((void **)&c->klist_z, sizeof(int) * (npz + 1)));
    RT(rtf_h2d(c->klist_z, pz.data(), sizeof(int) * (npz + 1), c->stream));
    RT(rtf_h2d(c->lev_ext_i, ni.data(), sizeof(int) * (npz + 1), c->stream));
    RT(rtf_h2d(c->lev_ext_d, nd.data(), sizeof(double) * (npz + 1), c->stream));
    RT(rtf_sync(c->stream));
  }
  RT(rtf_sync(c->stream));
  for (int x = 0; x < 2; x++) {
    fv3_ctx::LevSel &ls = c->lev_sel[x];
    std::vector<int> plain, damped;
    for (int k = 0; k < npz; k++) {
      const bool wd = lv->damp_w[k] > 1.E-5 && !(x == 1 && lv->nord_w[k] == 0);
      ((lv->damp_vt[k] > 1.E-4 || wd || lv->damp_t[k] > 1.E-4) ? damped : plain).push_back(k);
    }
    ls.n_plain = (int)plain.size();
    ls.n_damp = (int)damped.size();
    plain.insert(plain.end(), damped.begin(), damped.end());
    if (!ls.klist) RT(rt_malloc((void **)&ls.klist, sizeof(int) * npz));
    RT(rtf_h2d(ls.klist, plain.data(), sizeof(int) * npz, c->stream));
    std::vector<int> pm, rm;
    for (int k = 0; k < npz; k++) {
      const bool nk = lv->nord_k[k] == 1 || (x == 1 && lv->nord_k[k] == 0);
      ((nk && !(lv->damp_vt[k] > 1.E-5) && !(lv->d_con_k[k] > 1.E-5)) ? pm : rm).push_back(k);
    }
    ls.n_plain_m = (int)pm.size();
    ls.n_rest_m = (int)rm.size();
    pm.insert(pm.end(), rm.begin(), rm.end());
    ls.side_ok = (rm == damped);
    if (!ls.klist_m) RT(rt_malloc((void **)&ls.klist_m, sizeof(int) * npz));
    RT(rtf_h2d(ls.klist_m, pm.data(), sizeof(int) * npz, c->stream));
    RT(rtf_sync(c->stream));
  }
  lev_activate(c, 0);
  c->lev_max_nord = c->lev_max_nord_v = c->lev_max_nord_w = c->lev_max_nord_t = 0;
  c->lev_has_dcon = c->lev_has_vt_damp = c->lev_has_w_damp = c->lev_has_w_damp_hi = false;
  c->lev_has_damp_v4 = c->lev_has_damp_v5 = c->lev_has_damp_t = false;
  for (int k = 0; k < npz; k++) {
    c->lev_max_nord = std::max(c->lev_max_nord, lv->nord_k[k]);
    if (lv->damp_vt[k] > 1.E-5) { c->lev_has_damp_v5 = true; c->lev_max_nord_v = std::max(c->lev_max_nord_v, lv->nord_v[k]); }
    if (lv->damp_vt[k] > 1.E-4) c->lev_has_damp_v4 = true;
    if (lv->damp_t[k] > 1.E-4) { c->lev_has_damp_t = true; c->lev_max_nord_t = std::max(c->lev_max_nord_t, lv->nord_t[k]); }
    if (lv->damp_w[k] > 1.E-5) c->lev_max_nord_w = std::max(c->lev_max_nord_w, lv->nord_w[k]);
    if (lv->d_con_k[k] > 1.E-5) c->lev_has_dcon = true;
    if (lv->damp_vt[k] > 1.E-5 || lv->damp_t[k] > 1.E-4) c->lev_has_vt_damp = true;
    if (lv->damp_w[k] > 1.E-5) c->lev_has_w_damp = true;
    if (lv->damp_w[k] > 1.E-5 && lv->nord_w[k] > 0) c->lev_has_w_damp_hi = true;
  }
  c->lev_ready = true;
  return 0;
}

// ---- fv_tp_2d as a stand-alone kernel (unit-test surface; the fused d_sw kernels call the same
// tile routine directly) --------------------------------------------------------------------
template <int TI, int TJ>
struct Tp2dKernel {
  Grid g;
  const double *q, *crx, *cry, *xfx, *yfx, *ra_x, *ra_y, *mfx, *mfy, *mass;
  double *fx, *fy;
  int hord, nord;
  double damp_c;
  using TS = Tp2dScratch<TI, TJ>;
  using DS = DelnScratch<TI, TJ>;
  static constexpr int nQ = (TI + 6) * (TJ + 6);
  static constexpr int nScr = TS::total > DS::total ? TS::total : DS::total;
  static constexpr int nFXt = (TI + 1) * TJ, nFYt = TI * (TJ + 1);
  static constexpr int lds_doubles = 2 * nQ + nScr + nFXt + nFYt;

  FV3_HD void operator()(int bx, int by, int bz, int tid, double *lds) const {
    const int k = bz;
    const TileBox b = make_box<TI, TJ>(g, bx, by);
    const int i0 = b.i0, j0 = b.j0;
    const size_t oA = (size_t)k * g.nA(), oCX = (size_t)k * g.nCX(), oCY = (size_t)k * g.nCY();
    const size_t oFX = (size_t)k * g.nFX(), oFY = (size_t)k * g.nFY();
    double *p = lds;
    const Tile sq{p, i0 - 3, j0 - 3, TI + 6}; p += nQ;
    const Tile sm{p, i0 - 3, j0 - 3, TI + 6}; p += nQ;
    double *scr = p; p += nScr;
    const Tile sfx{p, i0, j0, TI + 1}; p += nFXt;
    const Tile sfy{p, i0, j0, TI}; p += nFYt;
    load_tile<TI + 6, TJ + 6>(sq, q + oA, g.nid, g.isd, g.ied, g.jsd, g.jed, tid);
    const bool damp_on = nord >= 0 && damp_c > 1.e-4;
    const bool use_mass = (mfx && mfy && mass);
    if (damp_on && use_mass) load_tile<TI + 6, TJ + 6>(sm, mass + oA, g.nid, g.isd, g.ied, g.jsd, g.jed, tid);
    FV3_SYNC();
    tp2d_tile<TI, TJ>(g, b, tid, sq, crx + oCX, cry + oCY, xfx + oCX, yfx + oCY,
                      ra_x ? ra_x + (size_t)k * g.nRX() : nullptr, ra_y ? ra_y + (size_t)k * g.nRY() : nullptr, hord,
                      scr, sfx, sfy);
    FV3_TILE_FOR((TI + 1), (nFXt) / (TI + 1), li_, lj_) {
      const int i = i0 + li_, j = j0 + lj_;
      if (i > b.ilast + 1 || j > b.jlast) continue;
      const double m = (mfx && mfy) ? mfx[oFX + g.iFX(i, j)] : xfx[oCX + g.iCX(i, j)];
      sfx(i, j) = sfx(i, j) * m;
    }
    FV3_TILE_FOR(TI, (nFYt) / TI, li_, lj_) {
      const int i = i0 + li_, j = j0 + lj_;
      if (i > b.ilast || j > b.jlast + 1) continue;
      const double m = (mfx && mfy) ? mfy[oFY + g.iFY(i, j)] : yfx[oCY + g.iCY(i, j)];
      sfy(i, j) = sfy(i, j) * m;
    }
    FV3_SYNC();
    // deln_flux: with mfx/mfy it needs mass as well (tp_core.F90:201); without, mass is never passed
    if (damp_on && ((mfx && mfy) ? (mass != nullptr) : true)) {
      const double damp = ipow(damp_c * g.da_min, nord + 1);
      Tile fxd, fyd;
      const bool wm = use_mass;
      deln_tile<TI, TJ>(g, b, tid, sq, nord, damp, !wm, scr, fxd, fyd);
      const double damp2 = 0.5 * damp;
      FV3_TILE_FOR((TI + 1), (nFXt) / (TI + 1), li_, lj_) {
        const int i = i0 + li_, j = j0 + lj_;
        if (i > b.ilast + 1 || j > b.jlast) continue;
        sfx(i, j) = wm ? sfx(i, j) + damp2 * (sm(i - 1, j) + sm(i, j)) * fxd(i, j) : sfx(i, j) + fxd(i, j);
      }
      FV3_TILE_FOR(TI, (nFYt) / TI, li_, lj_) {
        const int i = i0 + li_, j = j0 + lj_;
        if (i > b.ilast || j > b.jlast + 1) continue;
        sfy(i, j) = wm ? sfy(i, j) + damp2 * (sm(i, j - 1) + sm(i, j)) * fyd(i, j) : sfy(i, j) + fyd(i, j);
      }
      FV3_SYNC();
    }
    FV3_TILE_FOR((TI + 1), (nFXt) / (TI + 1), li_, lj_) {
      const int i = i0 + li_, j = j0 + lj_;
      if (i > b.ilast + 1 || j > b.jlast) continue;
      if (i == i0 + TI && i <= g.ie) continue;
      fx[oFX + g.iFX(i, j)] = sfx(i, j);
    }
    FV3_TILE_FOR(TI, (nFYt) / TI, li_, lj_) {
      const int i = i0 + li_, j = j0 + lj_;
      if (i > b.ilast || j > b.jlast + 1) continue;
      if (j == j0 + TJ && j <= g.je) continue;
      fy[oFY + g.iFY(i, j)] = sfy(i, j);
    }
  }
};

// ---- cubed sphere (grid_type < 3): pass kernels ------------------------------------------------------------------------
template <class F>
static int launch_box(fv3_ctx *c, const char *label, int i0, int i1, int j0, int j1, int nk, const F &f) {
  if (i1 < i0 || j1 < j0 || nk <= 0) return 0;
  static const int rows_env = [] {
    const char *e = std::getenv("FV3_MI355X_BOX_ROWS");
    const int n = e ? std::atoi(e) : 16;
    return (n == 4 || n == 8 || n == 16) ? n : 16;
  }();
  const int rows = (j1 - j0 + 1 >= 64) ? rows_env : 4;
  Dim3 grid;
  grid.x = (unsigned)((i1 - i0 + 64) / 64);
  grid.y = (unsigned)((j1 - j0 + rows) / rows);
  grid.z = (unsigned)nk;
  return launch_p(c, label, grid, 0, BoxPass<F>{i0, i1, j0, j1, f, rows});
}
// where a pass runs: the whole box (w = 0) or only the frame of width w along the face edges; all levels (klist = null, nk
// levels) or the nk levels of a device list
struct PassRegion {
  int w;
  const int *klist;
  int nk;
};
template <class F>
static int launch_pass(fv3_ctx *c, const char *label, int i0, int i1, int j0, int j1, const PassRegion &rg, const F &f) {
  if (i1 < i0 || j1 < j0 || rg.nk <= 0) return 0;
  const Grid &g = c->g;
  if (rg.w <= 0 || g.npy - rg.w <= rg.w + 1 || g.npx - rg.w <= rg.w + 1) {
    if (!rg.klist) return launch_box(c, label, i0, i1, j0, j1, rg.nk, f);
    Dim3 grid;
    grid.x = (unsigned)((i1 - i0 + 64) / 64);
    grid.y = (unsigned)((j1 - j0 + 4) / 4);
    grid.z = (unsigned)rg.nk;
    return launch_p(c, label, grid, 0, BoxPassK<F>{i0, i1, j0, j1, rg.klist, f});
  }
  const int js1 = rg.w < j1 ? rg.w : j1, jn0 = (g.npy - rg.w) > j0 ? (g.npy - rg.w) : j0;
  const int nsn = (js1 - j0 + 1) + (j1 - jn0 + 1), nmid = jn0 - js1 - 1;
  FramePass<F> kf{i0, i1, j0, j1, rg.w, g.npx, g.npy, (i1 - i0 + 64) / 64, (nsn + 3) / 4, rg.klist, f};
  const int iw1 = rg.w < i1 ? rg.w : i1, ie0 = (g.npx - rg.w) > i0 ? (g.npx - rg.w) : i0;
  kf.wc = (iw1 - i0 + 1 <= 8 && i1 - ie0 + 1 <= 8) ? 8 : 16;
  const int we_rows = 256 / kf.wc;
  Dim3 grid;
  grid.x = (unsigned)(kf.nbx * kf.nby_sn + (nmid > 0 ? 2 * ((nmid + we_rows - 1) / we_rows) : 0));
  grid.y = 1;
  grid.z = (unsigned)rg.nk;
  return launch_p(c, label, grid, 0, kf);
}
// n-th work array of the cubed-sphere kernels: (nid+1) x (njd+1) x (npz+1) doubles, allocated on first use
static double *cs_scratch(fv3_ctx *c, int n) {
  if (!c->cs_scr[n]) {
    if (rt_malloc((void **)&c->cs_scr[n], sizeof(double) * c->g.nB() * (size_t)(c->g.npz + 1))) return nullptr;
  }
  return c->cs_scr[n];
}
static bool is_cubed(const fv3_ctx *c) { return c->g.grid_type < 3; }

extern "C" int fv3_grid_upload_cubed(fv3_ctx *c, const fv3_grid_cubed *h) {
  if (!c || !h) return fail("fv3_grid_upload_cubed: null argument");
  if (!is_cubed(c)) return fail("fv3_grid_upload_cubed: the context is not a cubed-sphere face (grid_type < 3)");
  if (!h->edge_w || !h->edge_e || !h->edge_s || !h->edge_n || !h->rsina) return fail("fv3_grid_upload_cubed: null array");
  c->host_area.clear();
  c->global_area = 0.;
  const Grid &g = c->g;
  const size_t ne = (size_t)g.npx, nr = (size_t)(g.nx + 1) * (g.ny + 1);
  const size_t nr8 = (nr + 7) & ~(size_t)7;
  const size_t total = 4 * ((ne + 7) & ~(size_t)7) + nr8 + 4 * g.nA() + 3 * (2 * g.nA() + g.nFY() + g.nFX());
  if (!c->cg_dev) RT(rt_malloc((void **)&c->cg_dev, total * sizeof(double)));
  double *p = c->cg_dev;
  const double *src[4] = {h->edge_w, h->edge_e, h->edge_s, h->edge_n};
  const double **dst[4] = {&c->cg.edge_w, &c->cg.edge_e, &c->cg.edge_s, &c->cg.edge_n};
  for (int n = 0; n < 4; n++) {
    RT(rtf_h2d(p, src[n], ne * sizeof(double), c->stream));
    *dst[n] = p - 1;  // 1-based
    p += (ne + 7) & ~(size_t)7;
  }
  RT(rtf_h2d(p, h->rsina, nr * sizeof(double), c->stream));
  c->cg.rsina = p;
  p += nr8;
  c->cg.a11 = c->cg.a12 = c->cg.a21 = c->cg.a22 = nullptr;
  if (h->a11 && h->a12 && h->a21 && h->a22) {
    const double *am[4] = {h->a11, h->a12, h->a21, h->a22};
    const double **ad[4] = {&c->cg.a11, &c->cg.a12, &c->cg.a21, &c->cg.a22};
    for (int n = 0; n < 4; n++) {
      RT(rtf_h2d(p, am[n], g.nA() * sizeof(double), c->stream));
      *ad[n] = p;
      p += g.nA();
    }
  }
  c->cg.ec1 = c->cg.ec2 = c->cg.en1 = c->cg.en2 = nullptr;
  if (h->ec1 && h->ec2 && h->en1 && h->en2) {
    const double *am[4] = {h->ec1, h->ec2, h->en1, h->en2};
    const double **ad[4] = {&c->cg.ec1, &c->cg.ec2, &c->cg.en1, &c->cg.en2};
    const size_t sz[4] = {3 * g.nA(), 3 * g.nA(), 3 * g.nFY(), 3 * g.nFX()};
    for (int n = 0; n < 4; n++) {
      RT(rtf_h2d(p, am[n], sz[n] * sizeof(double), c->stream));
      *ad[n] = p;
      p += sz[n];
    }
  }
  for (int n = 0; n < 12; n++) c->cg.corner_f[n] = h->corner_f[n];
  RT(rtf_sync(c->stream));
  c->cg.ready = 1;
  return 0;
}

// fv_tp_2d on a cubed-sphere face: fx, fy = the fluxes of tp_core.F90:187-224 (times mfx / mfy or xfx / yfx); scratch 4..7
static int tp2d_cubed(fv3_ctx *c, int nk, const double *q, const double *crx, const double *cry, int hord, double *fx,
                      double *fy, const double *xfx, const double *yfx, const double *ra_x, const double *ra_y,
                      const double *mfx, const double *mfy, const char *label = "fv_tp_2d", const PassRegion *region = nullptr) {
  const Grid &g = c->g;
  const PassRegion rg = region ? *region : PassRegion{0, nullptr, nk};
  Tp2dCubedState s;
  s.g = g; s.q = q; s.crx = crx; s.cry = cry; s.xfx = xfx; s.yfx = yfx; s.ra_x = ra_x; s.ra_y = ra_y;
  s.mfx = mfx; s.mfy = mfy; s.fx = fx; s.fy = fy; s.hord = hord;
  double **scr[4] = {&s.fx2, &s.fy2, &s.q_i, &s.q_j};
  for (int n = 0; n < 4; n++)
    if (!(*scr[n] = cs_scratch(c, 4 + n))) return fail("fv_tp_2d: out of device memory");
  // Frame launches (rg.w = the consumer's frame wo + cubed_reach): the consumers (D4, D9, the zh / tracer updates) read the fluxes of
  // the faces of their own cells, i.e. within wo + 1 of an edge; an outer-sweep face reads q_i / q_j three cells further (wo + 4),
  // q_i / q_j the inner fluxes one face further (wo + 5).  Each pass runs on the frame it is read on, not on the widest one.
  PassRegion r2 = rg, r3 = rg;
  if (rg.w > 0 && c->cubed_reach >= 5) {
    r2.w = rg.w - (c->cubed_reach - 4);
    r3.w = rg.w - (c->cubed_reach - 1);
  }
  RT(launch_pass(c, label, g.isd, g.ied, g.jsd, g.jed, rg, Tp2dCubedT1{s}));
  RT(launch_pass(c, label, g.isd, g.ied, g.jsd, g.jed, r2, Tp2dCubedT2{s}));
  RT(launch_pass(c, label, g.is, g.ie + 1, g.js, g.je + 1, r3, Tp2dCubedT3{s}));
  return 0;
}

// the same on the frame of width w3 (flux points) for up to three fields in one launch (cubed_tpf.h); fields[0] is weighted with
// xfx / yfx, the following ones with fields[0]'s fluxes; FV3_MI355X_FRAME_FUSED=0 falls back to the passes
static int tp2d_frame_fused(fv3_ctx *c, const TpfField *fields, int nf, const double *crx, const double *cry, const double *xfx,
                            const double *yfx, int w3, const int *klist, int nk, const char *label, bool full = false,
                            const double *emfx = nullptr, const double *emfy = nullptr, const double *dfx = nullptr,
                            const double *dfy = nullptr, const double *dcoef = nullptr) {
  if (nk <= 0) return 0;
  const Grid &g = c->g;
  Tp2dFrameFused kf;
  kf.g = g;
  for (int n = 0; n < 3; n++) kf.f[n] = fields[n < nf ? n : 0];
  kf.nf = nf; kf.crx = crx; kf.cry = cry; kf.xfx = xfx; kf.yfx = yfx; kf.w3 = w3; kf.klist = klist;
  kf.full = full ? 1 : 0;
  kf.emfx = emfx; kf.emfy = emfy;
  kf.dfx = dfx; kf.dfy = dfy; kf.dcoef = dcoef;
  kf.nS = (g.nx + 1 + kTfT - 1) / kTfT;
  if (full) {            // the whole face (the levels the marching kernels do not take): bands of 5 rows
    kf.w3 = 7;
    kf.nW = (g.ny + 1 + kf.w3 - 1) / kf.w3;
  } else {
    const int nmid = (g.npy - w3 - 1) - (w3 + 1) + 1;
    kf.nW = (nmid + kTfT - 1) / kTfT;
  }
  Dim3 grid;
  grid.x = (unsigned)kf.ntiles(); grid.y = 1; grid.z = (unsigned)nk;
  return launch_p(c, label, grid, (size_t)kTfArrays * kTfMaxN, kf);
}
static bool deln_fused_on() {
  static const int v = [] {
    const char *e = std::getenv("FV3_MI355X_DELN_FUSED");
    return e ? std::atoi(e) : 1;
  }();
  return v != 0;
}
static bool flux_march_on() {
  static const int v = [] {
    const char *e = std::getenv("FV3_MI355X_FLUX_MARCH");
    return e ? std::atoi(e) : 1;
  }();
  return v != 0;
}
static bool frame_fused_on() {
  static const int v = [] {
    const char *e = std::getenv("FV3_MI355X_FRAME_FUSED");
    return e ? std::atoi(e) : 1;
  }();
  return v != 0;
}

static int csw_march(fv3_ctx *c, const CswArgs &ca);
static int csw_cubed(fv3_ctx *c, const CswArgs &ca) {
  const Grid &g = c->g;
  double *scr[4];
  for (int n = 0; n < 4; n++)
    if (!(scr[n] = cs_scratch(c, n))) return fail("c_sw: out of device memory");
  CswCubedState s = make_csw_cubed(g, ca, scr);
  const int npz = g.npz;
  // Hybrid (see dsw_cubed): d2a2c_vect switches to its edge forms within npt = 4 points of a face edge, so the frame the
  // passes own is wider than in d_sw.  The passes run FIRST (P3 leaves the interpolated uc, vc on the wider frame of the
  // intermediates, which P4 / P5 read), then the marching kernel writes the points it owns -- the divergence too, in the
  // non-orthogonal form of the cubed sphere --, then the divergence of the frame, which reads the final ua, va, by the pass.
  const int wo = c->cubed_frame_c, wm = wo + c->cubed_reach;
  const bool hyb = c->use_march && wo > 0 && g.npx == g.npy && g.npx - 1 >= 2 * wm + 8;
  const PassRegion rm{hyb ? wm : 0, nullptr, npz}, ro{hyb ? wo : 0, nullptr, npz};
  s.divg = hyb ? 0 : 1;
  // Two lanes (round 4, see dsw_cubed): the frame passes on the side stream BESIDE the marching kernel.  The passes keep what they
  // read back of ua, va, uc, vc, ut, vt in work copies (scratch 8 .. 13: d_sw's, free here) and write an output only where they own
  // it (CswCubedState::wr), so every output point has one writer and the marching kernel's points are not read by them.  The
  // divergence of the frame reads the final ua, va of both: after the join, from the outputs.
  const bool lanes = hyb && lanes_pay(c);
  if (lanes) {
    double **wk[6] = {&s.ua_w, &s.va_w, &s.uc_w, &s.vc_w, &s.ut_w, &s.vt_w};
    for (int n = 0; n < 6; n++)
      if (!(*wk[n] = cs_scratch(c, 8 + n))) return fail("c_sw: out of device memory");
    s.own_w = wo;
    RT(lane_prepare(c));
    RT(lane_op(c, kLaneFork));
    CswArgs cm = ca;
    cm.mask_w = wo;
    RT(csw_march(c, cm));
    c->lane = 1;
  }
  auto passes = [&]() -> int {
    RT(launch_pass(c, "cswc_p1", g.isd, g.ied, g.jsd, g.jed, rm, CswCubedP1{s}));
    RT(launch_pass(c, "cswc_p2", g.is - 2, g.ie + 2, g.js - 2, g.je + 2, rm, CswCubedP2{s}));
    RT(launch_box(c, "cswc_p2c", 0, 2, 0, 0, npz, CswCubedP2c{s}));
    RT(launch_pass(c, "cswc_p3", g.is - 1, g.ie + 2, g.js - 1, g.je + 2, rm, CswCubedP3{s}));
    RT(launch_pass(c, "cswc_p4", g.is - 1, g.ie + 1, g.js - 1, g.je + 1, rm, CswCubedP4{s}));
    s.own_w = hyb ? wo : 0;
    RT(launch_pass(c, "cswc_p5", g.is - 1, g.ie + 1, g.js - 1, g.je + 1, ro, CswCubedP5{s}));
    return 0;
  };
  const int rc = passes();
  if (lanes) {
    c->lane = 0;
    // (also after a failed launch: what the side stream already holds must be ordered before whatever the caller issues next)
    const int rj = lane_op(c, kLaneJoin);
    if (rc) return rc;
    RT(rj);
  } else {
    if (rc) return rc;
    if (hyb) {
      CswArgs cm = ca;
      cm.mask_w = wo;
      RT(csw_march(c, cm));
    }
  }
  if (hyb && ca.nord > 0) {  // the marching kernel formed the divergence of the points it owns; the frame by the pass
    s.divg = 2;
    s.ua_w = s.va_w = s.uc_w = s.vc_w = s.ut_w = s.vt_w = nullptr;
    RT(launch_pass(c, "cswc_div", g.is, g.ie + 1, g.js, g.je + 1, ro, CswCubedP3{s}));
  }
  return 0;
}

static int need_trc(fv3_ctx *c);
extern "C" int fv3_fv_tp_2d(fv3_ctx *c, int nk, const double *q, const double *crx, const double *cry, int hord,
                            double *fx, double *fy, const double *xfx, const double *yfx, const double *ra_x,
                            const double *ra_y, const double *mfx, const double *mfy, const double *mass, int nord,
                            double damp_c) {
  if (!c || !c->grid_ready) return fail("fv3_fv_tp_2d: context has no grid (call fv3_grid_upload)");
  if (!tp_ord_supported_tr(hord)) return fail("fv3_fv_tp_2d: hord=%d not supported (5,-5,6,7,8,9,10,11,12,13)", hord);
  if (nord > 2) return fail("fv3_fv_tp_2d: nord=%d > 2", nord);
  if ((mfx == nullptr) != (mfy == nullptr)) return fail("fv3_fv_tp_2d: mfx and mfy must be given together");
  if (is_cubed(c)) {
    if (nk > c->g.npz + 1) return fail("fv3_fv_tp_2d: nk > npz + 1 on a cubed-sphere context");
    if (tp2d_cubed(c, nk, q, crx, cry, hord, fx, fy, xfx, yfx, ra_x, ra_y, mfx, mfy)) return 1;
    if (nord >= 0 && damp_c > 1.e-4 && !(mfx && !mass)) {  // deln_flux (tp_core.F90:227-239): cubed_damp.h, one order for all levels
      if (need_trc(c)) return 1;
      const Grid &g = c->g;
      if (nk > g.npz) return fail("fv3_fv_tp_2d: deln_flux damping on a cubed-sphere context takes at most npz levels");
      std::vector<int> ni(g.npz, nord);
      std::vector<double> cd(g.npz, damp_c);
      RT(rtf_h2d(c->trc_i + g.npz, ni.data(), sizeof(int) * g.npz, c->stream));
      RT(rtf_h2d(c->trc_d + 2 * g.npz, cd.data(), sizeof(double) * g.npz, c->stream));
      RT(rtf_sync(c->stream));
      DelnCubedState d;
      d.g = g; d.q = q; d.mass = mfx ? mass : nullptr; d.fx = fx; d.fy = fy; d.nord = c->trc_i + g.npz; d.coef = c->trc_d + 2 * g.npz; d.thresh = 1.E-4;
      d.corner_area = 0;
      d.d2 = cs_scratch(c, 4); d.fx2 = cs_scratch(c, 5); d.fy2 = cs_scratch(c, 6);
      if (!d.d2 || !d.fx2 || !d.fy2) return fail("fv3_fv_tp_2d: out of device memory");
      const PassRegion r{0, nullptr, nk};
      RT(launch_pass(c, "fv_tp_2d", g.isd, g.ied, g.jsd, g.jed, r, DelnCubedL1{d}));
      RT(launch_pass(c, "fv_tp_2d", g.isd, g.ied + 1, g.jsd, g.jed + 1, r, DelnCubedL24{d, 1, 0}));
      for (int n = 1; n <= nord; n++) {
        RT(launch_pass(c, "fv_tp_2d", g.isd, g.ied, g.jsd, g.jed, r, DelnCubedL3{d, n}));
        RT(launch_pass(c, "fv_tp_2d", g.isd, g.ied + 1, g.jsd, g.jed + 1, r, DelnCubedL24{d, 0, n}));
      }
      RT(launch_pass(c, "fv_tp_2d", g.is, g.ie + 1, g.js, g.je + 1, r, DelnCubedL5{d}));
    }
    return 0;
  }
  constexpr int TI = FV3_DSW_TI, TJ = FV3_DSW_TJ;
  Tp2dKernel<TI, TJ> kf{c->g, q, crx, cry, xfx, yfx, ra_x, ra_y, mfx, mfy, mass, fx, fy, hord, nord, damp_c};
  Dim3 grid;
  grid.x = (unsigned)((c->g.nx + TI - 1) / TI);
  grid.y = (unsigned)((c->g.ny + TJ - 1) / TJ);
  grid.z = (unsigned)nk;
  RT(launch_p(c, "fv_tp_2d", grid, Tp2dKernel<TI, TJ>::lds_doubles, kf));
  return 0;
}

// one thread per face of a line through ppm_face_tp, the 1-D operator of the LDS-tile kernels (ppm.h)
struct PpmLineTile {
  const double *h, *c;
  double *flux;
  int n, iord;
  double lim_fac;
  FV3_D void operator()(int bx, int, int, int tid, double *) const {
    for (int f = bx * kNT + tid; f <= n; f += kNT) flux[f] = ppm_face_tp(h + f + 3, 1, c[f], iord, lim_fac);   // face f + 1 between cells f, f + 1
  }
};

extern "C" int fv3_ppm_line(fv3_ctx *c, int iord, int which, const double *h, const double *cr, double *flux, int n) {
  if (!c || !c->grid_ready) return fail("fv3_ppm_line: context has no grid");
  if (!h || !cr || !flux || n < 1) return fail("fv3_ppm_line: bad arguments");
  if (which == 0) {
    if (!tp_ord_supported_tr(iord)) return fail("fv3_ppm_line: iord=%d not supported", iord);
    PpmLineTile kf{h, cr, flux, n, iord, c->g.lim_fac};
    RT(launch_p(c, "ppm_line", Dim3{1, 1, 1}, 1, kf));
    return 0;
  }
  if (which != 1 && which != 2) return fail("fv3_ppm_line: which = 0 (tile operator), 1 (marching, along the lanes), 2 (marching, register window)");
  if (which == 1 && n + 6 > 64) return fail("fv3_ppm_line: a line along the lanes holds at most 58 cells");
  switch (iord) {
    case 5: RT(launch_w(c, "ppm_line", 1, PpmLineMarch<5>{h, cr, flux, n, which - 1})); break;
    case -5: RT(launch_w(c, "ppm_line", 1, PpmLineMarch<-5>{h, cr, flux, n, which - 1})); break;
    case 6: RT(launch_w(c, "ppm_line", 1, PpmLineMarch<6>{h, cr, flux, n, which - 1})); break;
    case 8: RT(launch_w(c, "ppm_line", 1, PpmLineMarch<8>{h, cr, flux, n, which - 1})); break;
    case 10: RT(launch_w(c, "ppm_line", 1, PpmLineMarch<10>{h, cr, flux, n, which - 1})); break;
    default: return fail("fv3_ppm_line: the marching operators are built for iord 5, -5, 6, 8, 10 (what d_sw and update_dz_d take)");
  }
  return 0;
}

// Rows per wavefront segment: the configured value, shortened on small domains so that a launch still has a few
// thousand wavefronts (a wavefront marches tj + 6 rows one after the other: with too few of them the launch time is that
// serial march, not throughput).  Never below 8 rows (the 6 warm-up rows of every segment are overhead).
// FV3_MI355X_DEBUG_SEGMENTS=1: the segmentation every marching launch of the pair ended up with, on stderr
static void seg_report(const char *who, const MarchDims &d, int nlev) {
  static const int on = [] { const char *e = std::getenv("FV3_MI355X_DEBUG_SEGMENTS"); return e ? std::atoi(e) : 0; }();
  if (on)
    std::fprintf(stderr, "[fv3 segments] %s: %d level slots x %d strips x %d segments of %d rows; %d of the slots (spread evenly): %d segments of %d rows\n", who, nlev,
                 d.nstrips, d.nsegs, d.tj, d.alt_nk, d.alt_ng, d.alt_tj);
}
static int seg_rows(const fv3_ctx *c, int tj_conf, int nlev_slots) {
  const Grid &g = c->g;
  if (c->tj_fixed) return tj_conf;
  const int nstrips = num_strips(g);
  const long have = (long)nstrips * (nlev_slots > 0 ? nlev_slots : 1);
  const int want_segs = (int)((2048 + have - 1) / have);
  int tj = (g.ny + want_segs - 1) / want_segs;
  if (tj < 8) tj = 8;
  return tj < tj_conf ? tj : tj_conf;
}

// geometry mode (Grid::geom) as a compile-time constant
template <class F>
static int dispatch_geom(int geom, F &&f) {
  switch (geom) {
    case 2: return f(std::integral_constant<int, 2>{});
    case 1: return f(std::integral_constant<int, 1>{});
    default: return f(std::integral_constant<int, 0>{});
  }
}

extern "C" int fv3_c_sw(fv3_ctx *c, double *delpc, const double *delp, double *ptc, const double *pt,
                        const double *u, const double *v, const double *w, double *uc, double *vc, double *ua,
                        double *va, double *wc, double *ut, double *vt, double *divg_d, int nord, double dt2,
                        int hydrostatic, int dord4) {
  (void)dord4;  // ua, va are produced on is-1:ie+1 (what c_sw/d_sw read); see header
  if (!c || !c->grid_ready) return fail("fv3_c_sw: context has no grid (call fv3_grid_upload)");
  if (!hydrostatic && (!w || !wc)) return fail("fv3_c_sw: nonhydrostatic call needs w and wc");
  if (is_cubed(c)) {
    if (!c->cg.ready) return fail("fv3_c_sw: cubed-sphere context without fv3_grid_upload_cubed");
    return csw_cubed(c, CswArgs{delpc, ptc, wc, uc, vc, ua, va, ut, vt, divg_d, delp, pt, u, v, w, nord, hydrostatic, dt2});
  }
  if (c->use_march) return csw_march(c, CswArgs{delpc, ptc, wc, uc, vc, ua, va, ut, vt, divg_d, delp, pt, u, v, w, nord, hydrostatic, dt2});
  constexpr int TI = FV3_CSW_TI, TJ = FV3_CSW_TJ;
  CswTile<TI, TJ> kf;
  kf.g = c->g;
  kf.a = CswArgs{delpc, ptc, wc, uc, vc, ua, va, ut, vt, divg_d, delp, pt, u, v, w, nord, hydrostatic, dt2};
  Dim3 grid;
  CswTile<TI, TJ>::grid_dims(c->g, grid.x, grid.y);
  grid.z = (unsigned)c->g.npz;
  RT(launch_p(c, "c_sw", grid, CswTile<TI, TJ>::lds_doubles, kf));
  return 0;
}

static int csw_march(fv3_ctx *c, const CswArgs &ca) {
  {
    // rows per segment: 64 for the two-levels-per-wavefront kernel (one wavefront per SIMD); the uniform-metric kernel
    // (one level per wavefront, four per SIMD, bandwidth-bound) does better with many short segments (measured 16-40: 24)
    const int tj_csw = c->march_tj_csw ? c->march_tj_csw : (c->g.geom == 2 ? 24 : 64);
    MarchDims md = make_csw_dims(c->g, seg_rows(c, tj_csw, c->g.npz));
    // uniform metrics: nothing to share between levels, one level per wavefront at four wavefronts per SIMD is faster
    int kpw = c->csw_kpw ? c->csw_kpw : (c->g.geom == 2 ? 1 : 2);
    const int nkg = (c->g.npz + kpw - 1) / kpw;
    if (c->g.geom == 2 && kpw == 1 && ca.mask_w == 0)   // whole rounds of the chip at four wavefronts per SIMD
      balance_segments(md, nkg, c->g.ny + 4, 4 * c->round_simds, md.tj);
    const int nw = md.nwaves(nkg);
    seg_report("c_sw", md, nkg);
    if (ca.mask_w > 0) {  // the interior of a cubed-sphere face: general metrics + the cubed switches
      if (kpw == 1) return launch_w(c, "c_sw", nw, CswMarch<1, 0, true>{c->g, ca, md, nkg});
      return launch_w(c, "c_sw", nw, CswMarch<2, 0, true>{c->g, ca, md, nkg});
    }
    auto go = [&](auto GMc) -> int {
      constexpr int GM = decltype(GMc)::value;
      if (kpw == 2) return launch_w(c, "c_sw", nw, CswMarch<2, GM>{c->g, ca, md, nkg});
      return launch_w(c, "c_sw", nw, CswMarch<1, GM>{c->g, ca, md, nkg});
    };
    return dispatch_geom(c->g.geom, go);
  }
}

// compile-time scheme dispatch for the marching kernels
template <class Fn>
static int dispatch_hord(int hord, Fn &&fn) {
  switch (hord) {
    case 5: return fn(std::integral_constant<int, 5>());
    case -5: return fn(std::integral_constant<int, -5>());
    case 6: return fn(std::integral_constant<int, 6>());
    case 8: return fn(std::integral_constant<int, 8>());
    case 10: return fn(std::integral_constant<int, 10>());
  }
  return fail("unsupported hord %d", hord);
}

// tracer_2d also takes hord_tr = 9 / 13 (the same scheme), 11, 12: instantiated for the tracer kernels only
template <class Fn>
static int dispatch_hord_tr(int hord, Fn &&fn) {
  switch (hord) {
    case 7: return fn(std::integral_constant<int, 7>());
    case 9:
    case 13: return fn(std::integral_constant<int, 9>());
    case 11: return fn(std::integral_constant<int, 11>());
    case 12: return fn(std::integral_constant<int, 12>());
  }
  return dispatch_hord(hord, fn);
}

static int ensure_mflux(fv3_ctx *c) {
  const Grid &g = c->g;
  if (!c->mflux[0]) RT(rt_malloc((void **)&c->mflux[0], sizeof(double) * g.nFX() * g.npz));
  if (!c->mflux[1]) RT(rt_malloc((void **)&c->mflux[1], sizeof(double) * g.nFY() * g.npz));
  return 0;
}

// d_sw transports on the wave-marching fv_tp_2d (dsw_march.h)
// region (fused kernel only): 0 = every strip / segment, 1 = those that do not touch the halo (interior box),
// 2 = the frame around the interior box
static bool dsw_has_interior(const fv3_ctx *c) {
  const MarchDims mf = make_march_dims(c->g, seg_rows(c, c->march_tj_fused, c->g.npz));
  // A strip's lanes reach 3 columns past the cells it owns and a segment 3 rows past its last row, so strip NS-2 /
  // segment NG-2 stay clear of the halo only if the ragged last strip owns >= 3 cells and the last segment >= 3 rows;
  // otherwise the whole domain is left to the 'rest' phase (no interior launch before the exchange has completed).
  const int last_cols = c->g.nx - kStripCells * (mf.nstrips - 1), last_rows = c->g.ny - mf.tj * (mf.nsegs - 1);
  return mf.nstrips >= 3 && mf.nsegs >= 3 && last_cols >= 3 && last_rows >= 3;
}

static int dsw_transport_march(fv3_ctx *c, const DswArgs &a, int region = 0) {
  const Grid &g = c->g;
  if (ensure_mflux(c)) return 1;
  MarchDims md = make_march_dims(g, seg_rows(c, c->march_tj, g.npz));
  md.klist = c->klist;
  const int nw = md.nwaves(c->n_plain);
  if (c->use_fused && !a.use_cond && a.hord_dp == a.hord_tm && (a.hydrostatic || a.hord_dp == a.hord_vt)) {
    if (c->n_plain == 0) return 0;
    // (the interior of a cubed-sphere face, six faces a launch: seven 55-row segments measured 3 % better than eight of 48 there)
    MarchDims mf = make_march_dims(g, seg_rows(c, (a.mask_w && !c->tj_env_fused) ? 55 : c->march_tj_fused, g.npz));
    mf.klist = c->klist;
    const int NS = mf.nstrips, NG = mf.nsegs;
    auto box = [&](int s0, int ns, int g0, int ng) -> int {
      if (ns <= 0 || ng <= 0) return 0;
      mf.set_box(s0, ns, g0, ng);
      // the whole grid in one launch of the branch-free kernel (two wavefronts per SIMD): whole rounds of the chip
      if (FV3_BF && s0 == 0 && ns == NS && g0 == 0 && ng == NG && a.mask_w == 0)
        balance_segments(mf, c->n_plain, g.ny, 2 * c->round_simds, mf.tj);
      const int nwf = mf.nwaves(c->n_plain);
      seg_report("d_sw_fused", mf, c->n_plain);
      return dispatch_hord(a.hord_dp, [&](auto H) {
        constexpr int HORD = decltype(H)::value;
        if (g.geom == 2) {
          if (a.hydrostatic) return launch_w(c, "d_sw_fused", nwf, DswTransportFused<HORD, false, true, 2>{g, a, mf});
          return launch_w(c, "d_sw_fused", nwf, DswTransportFused<HORD, true, true, 2>{g, a, mf});
        }
        if (a.hydrostatic) return launch_w(c, "d_sw_fused", nwf, DswTransportFused<HORD, false, true>{g, a, mf});
        return launch_w(c, "d_sw_fused", nwf, DswTransportFused<HORD, true, true>{g, a, mf});
      });
    };
    if (region == 0 || !dsw_has_interior(c)) return region == 1 ? 0 : box(0, NS, 0, NG);
    if (region == 1) return box(1, NS - 2, 1, NG - 2);
    mf.set_frame();                            // south / north rows and west / east columns in one launch
    const int nwf = mf.nwaves(c->n_plain);
    return dispatch_hord(a.hord_dp, [&](auto H) {
      constexpr int HORD = decltype(H)::value;
      if (g.geom == 2) {
        if (a.hydrostatic) return launch_w(c, "d_sw_fused", nwf, DswTransportFused<HORD, false, true, 2>{g, a, mf});
        return launch_w(c, "d_sw_fused", nwf, DswTransportFused<HORD, true, true, 2>{g, a, mf});
      }
      if (a.hydrostatic) return launch_w(c, "d_sw_fused", nwf, DswTransportFused<HORD, false, true>{g, a, mf});
      return launch_w(c, "d_sw_fused", nwf, DswTransportFused<HORD, true, true>{g, a, mf});
    });
  }
  if (region == 1) return 0;  // the per-field kernels are not split
  double *fxs = c->mflux[0], *fys = c->mflux[1];
  int rc = dispatch_hord(a.hord_dp, [&](auto H) {
    DswDelpMarch<decltype(H)::value> kf{g, a, md, fxs, fys, 1};
    return launch_w(c, "d_sw_delp", nw, kf);
  });
  if (rc) return rc;
  auto scalar = [&](const char *label, int hord, const double *q, double *q_out) {
    return dispatch_hord(hord, [&](auto H) {
      DswScalarMarch<decltype(H)::value> kf{g, a, md, fxs, fys, q, q_out};
      return launch_w(c, label, nw, kf);
    });
  };
  if (!a.hydrostatic && (rc = scalar("d_sw_w", a.hord_vt, a.w, a.w_out))) return rc;
  if (a.use_cond && (rc = scalar("d_sw_qcon", a.hord_dp, a.q_con, a.q_con_out))) return rc;
  return scalar("d_sw_pt", a.hord_tm, a.pt, a.pt_out);
}

// d_sw momentum on the marching stencils (dsw_march.h) for the levels in klist_m[0 : n_plain_m]
// part = 0: everything; 1: only the KE / damping kernel (unfused path); 2: only the vorticity kernel
static int dsw_momentum_march(fv3_ctx *c, const DswArgs &a, int part = 0) {
  const Grid &g = c->g;
  const bool fused_m = c->use_fused != 0;
  if (!fused_m && !c->ke_scr) RT(rt_malloc((void **)&c->ke_scr, sizeof(double) * g.nB() * g.npz));
  if (fused_m) {
    MarchDims mf = make_march_dims(g, seg_rows(c, (a.mask_w && !c->tj_env_fused) ? 55 : c->march_tj_mom, g.npz));
    mf.klist = c->klist_m;
    if (FV3_BF && a.mask_w == 0)
      balance_segments(mf, c->n_plain_m, g.ny, ((g.geom == 2 && FV3_MOM_3W) ? 3 : 2) * c->round_simds, mf.tj);
    const int nwf = mf.nwaves(c->n_plain_m);
    seg_report("d_sw_mom_fused", mf, c->n_plain_m);
    return dispatch_hord(a.hord_vt, [&](auto H) {
      constexpr int HORD = decltype(H)::value;
      if (g.geom == 2) {
        switch (sw_class(a.hord_mt)) {
          case 5: return launch_w(c, "d_sw_mom_fused", nwf, DswMomentumFused<5, HORD, 2>{g, a, mf});
          case 6: return launch_w(c, "d_sw_mom_fused", nwf, DswMomentumFused<6, HORD, 2>{g, a, mf});
          default: return launch_w(c, "d_sw_mom_fused", nwf, DswMomentumFused<8, HORD, 2>{g, a, mf});
        }
      }
      if (a.rsina) {  // the interior of a cubed-sphere face
        switch (sw_class_cubed(a.hord_mt)) {
          case 5: return launch_w(c, "d_sw_mom_fused", nwf, DswMomentumFused<5, HORD, 0, true>{g, a, mf});
          case 6: return launch_w(c, "d_sw_mom_fused", nwf, DswMomentumFused<6, HORD, 0, true>{g, a, mf});
          case 108: return launch_w(c, "d_sw_mom_fused", nwf, DswMomentumFused<108, HORD, 0, true>{g, a, mf});
          case 110: return launch_w(c, "d_sw_mom_fused", nwf, DswMomentumFused<110, HORD, 0, true>{g, a, mf});
          case 111: return launch_w(c, "d_sw_mom_fused", nwf, DswMomentumFused<111, HORD, 0, true>{g, a, mf});
          default: return launch_w(c, "d_sw_mom_fused", nwf, DswMomentumFused<8, HORD, 0, true>{g, a, mf});
        }
      }
      switch (sw_class(a.hord_mt)) {
        case 5: return launch_w(c, "d_sw_mom_fused", nwf, DswMomentumFused<5, HORD>{g, a, mf});
        case 6: return launch_w(c, "d_sw_mom_fused", nwf, DswMomentumFused<6, HORD>{g, a, mf});
        default: return launch_w(c, "d_sw_mom_fused", nwf, DswMomentumFused<8, HORD>{g, a, mf});
      }
    });
  }
  if (part != 2) {
    MarchDims mk = make_march_dims(g, seg_rows(c, c->march_tj_ke, g.npz));
    mk.klist = c->klist_m;
    const int nwk = mk.nwaves(c->n_plain_m);
    int rc;
    switch (sw_class(a.hord_mt)) {
      case 5: rc = launch_w(c, "d_sw_ke", nwk, DswKeMarch<5>{g, a, mk, c->ke_scr}); break;
      case 6: rc = launch_w(c, "d_sw_ke", nwk, DswKeMarch<6>{g, a, mk, c->ke_scr}); break;
      default: rc = launch_w(c, "d_sw_ke", nwk, DswKeMarch<8>{g, a, mk, c->ke_scr}); break;
    }
    if (rc || part == 1) return rc;
  }
  MarchDims md = make_march_dims(g, seg_rows(c, c->march_tj, g.npz));
  md.klist = c->klist_m;
  const int nw = md.nwaves(c->n_plain_m);
  const double *ke = c->ke_scr;
  return dispatch_hord(a.hord_vt, [&](auto H) {
    DswVortMarch<decltype(H)::value> kf{g, a, md, ke};
    return launch_w(c, "d_sw_vort", nw, kf);
  });
}

template <int TI, int TJ>
static int run_a2b(fv3_ctx *c, const A2BCorners<TI, TJ> &kf, int nlev_max, const char *who = nullptr);
// d_sw on a cubed-sphere face (cubed_dsw.h); scratch 8..20
static int dsw_cubed(fv3_ctx *c, const DswArgs &a) {
  const Grid &g = c->g;
  if (!c->cg.ready) return fail("fv3_d_sw: cubed-sphere context without fv3_grid_upload_cubed");
  DswCubedState s;
  s.g = g; s.cg = c->cg; s.a = a; s.own_w = 0;
  double **scr[13] = {&s.ut, &s.vt, &s.fx, &s.fy, &s.gxw, &s.gyw, &s.gx, &s.gy, &s.ke, &s.wk, &s.dd, &s.svc, &s.suc};
  for (int n = 0; n < 13; n++)
    if (!(*scr[n] = cs_scratch(c, 8 + n))) return fail("d_sw: out of device memory");
  const int npz = g.npz, npx = g.npx, npy = g.npy;
  const char *L = "dswc_damp";
  // del-2n damping (cubed_damp.h): the passes of one operator; work arrays = scratch 4..6 of fv_tp_2d (free between its calls)
  auto deln = [&](const double *q, const double *mass, double *fx, double *fy, const int *nord, const double *coef, double thresh,
                  int corner_area, int nmax, double *out_fx2, double *out_fy2, const PassRegion &rk, int d2_slot = 4) -> int {
    DelnCubedState d;
    d.g = g; d.q = q; d.mass = mass; d.fx = fx; d.fy = fy; d.nord = nord; d.coef = coef; d.thresh = thresh; d.corner_area = corner_area;
    d.d2 = cs_scratch(c, d2_slot);
    d.fx2 = out_fx2 ? out_fx2 : cs_scratch(c, 5);
    d.fy2 = out_fy2 ? out_fy2 : cs_scratch(c, 6);
    if (!d.d2 || !d.fx2 || !d.fy2) return fail("d_sw: out of device memory");
    // away from the face corners: the chain in one LDS-tile launch (cubed_damp.h DelnFused); the passes keep the four corner
    // squares of 5 flux points (what the corner maps of copy_corners can reach) and the rim of 3 their intermediates need
    const int wo_d = 5, wm_d = wo_d + 3;
    const bool fused_d = deln_fused_on() && nmax <= DelnFused::kMaxN && g.npx == g.npy && g.npx - 1 >= 2 * wm_d + 8;
    const PassRegion r{0, rk.klist, rk.nk};
    auto pass = [&](int i0, int i1, int j0, int j1, int w, auto f) -> int {   // whole box, or its corner squares of side w
      if (!fused_d) return launch_pass(c, "dswc_deln", i0, i1, j0, j1, r, f);
      Dim3 gr;
      gr.x = 4; gr.y = 1; gr.z = (unsigned)rk.nk;
      return launch_p(c, "dswc_deln", gr, 0, CornerPass<decltype(f)>{i0, i1, j0, j1, w, g.npx, g.npy, rk.klist, f});
    };
    auto fused_launch = [&]() -> int {
      if (!fused_d) return 0;
      DelnFused kf{d, wo_d, rk.klist, fx ? 0 : 1};
      Dim3 gr;
      gr.x = (unsigned)((g.nx + 1 + DelnFused::TI - 1) / DelnFused::TI);
      gr.y = (unsigned)((g.ny + 1 + DelnFused::TJ - 1) / DelnFused::TJ);
      gr.z = (unsigned)rk.nk;
      return launch_p(c, "dswc_deln", gr, DelnFused::lds_doubles, kf);
    };
    RT(pass(g.isd, g.ied, g.jsd, g.jed, wm_d, DelnCubedL1{d}));
    RT(pass(g.isd, g.ied + 1, g.jsd, g.jed + 1, wm_d, DelnCubedL24{d, 1, 0}));
    for (int n = 1; n <= nmax; n++) {
      RT(pass(g.isd, g.ied, g.jsd, g.jed, wm_d, DelnCubedL3{d, n}));
      RT(pass(g.isd, g.ied + 1, g.jsd, g.jed + 1, wm_d, DelnCubedL24{d, 0, n}));
    }
    if (fx) RT(pass(g.is, g.ie + 1, g.js, g.je + 1, wo_d, DelnCubedL5{d}));
    RT(fused_launch());
    return 0;
  };
  if (!a.hydrostatic && c->lev_has_w_damp_hi) {
    if (!(s.wfx2 = cs_scratch(c, 25)) || !(s.wfy2 = cs_scratch(c, 26))) return fail("d_sw: out of device memory");
  }
  if (c->lev_has_damp_v5) {
    if (!(s.dfx2 = cs_scratch(c, 22)) || !(s.dfy2 = cs_scratch(c, 23))) return fail("d_sw: out of device memory");
  }
  const bool heat_pass = c->lev_has_dcon || g.do_diss_est;   // :1462, :1523
  if (heat_pass) {
    if (!(s.vortv = cs_scratch(c, 21))) return fail("d_sw: out of device memory");
  }
  if (!(a.dddmp < 1.E-5)) {
    if (!(s.smag = cs_scratch(c, 24))) return fail("d_sw: out of device memory");
  }
  if (a.use_cond) {
    if (!(s.gxq = cs_scratch(c, 27)) || !(s.gyq = cs_scratch(c, 28))) return fail("d_sw: out of device memory");
  }
  // contravariant winds of the whole face, all levels
  static const int d1_rows = [] { const char *e = std::getenv("FV3_MI355X_D1_ROWS"); return e ? std::atoi(e) : 1; }();
  if (d1_rows) {   // the marching form (cubed_dsw.h DswCubedD1aRows); FV3_MI355X_D1_ROWS=0: the point-wise pass
    Dim3 gr;
    gr.x = (unsigned)((g.ied + 1 - g.isd + 64) / 64);
    gr.y = (unsigned)((g.jed + 1 - g.jsd + 4 * DswCubedD1aRows::kRows) / (4 * DswCubedD1aRows::kRows));
    gr.z = (unsigned)npz;
    RT(launch_p(c, "dswc_d1", gr, 0, DswCubedD1aRows{s}));
  } else {
    RT(launch_box(c, "dswc_d1", g.isd, g.ied + 1, g.jsd, g.jed + 1, npz, DswCubedD1a{s}));
  }
  RT(launch_box(c, "dswc_d1b", 0, npx, 0, npy, npz, DswCubedD1b{s}));
  RT(launch_box(c, "dswc_d1c", 0, 3, 0, 0, npz, DswCubedD1c{s}));

  // Hybrid: away from the face edges the cubed-sphere d_sw is the general-metric stencil the marching kernels compute (with
  // the contravariant winds above in the place of uc, vc and the non-orthogonal B-grid winds of the kinetic energy), so those
  // kernels take the whole face and leave a frame of wo points along the edges to the passes; the passes' intermediates are
  // formed on a frame wider by the reach of the pass chain.  Levels the marching kernels do not take (sponge-level damping,
  // nord_k /= 1) go through the passes on the whole face.
  const int wo = c->cubed_frame, wm = wo + c->cubed_reach;
  const bool fits = wo > 0 && npx - 1 >= 2 * wm + 8 && npx == npy;
  const bool fused_ok = c->use_march && c->use_fused && !a.use_cond && a.hord_dp == a.hord_tm && (a.hydrostatic || a.hord_dp == a.hord_vt);
  const bool hyb_t = fits && fused_ok && c->n_plain > 0;
  const bool hyb_m = fits && c->use_march && c->use_fused && c->n_plain_m > 0 && a.dddmp < 1.E-5 && !g.do_diss_est;

  // the Courant numbers / area fluxes the passes read: d_sw's own arrays, or (two lanes) the frame's copies
  const double *cn_crx = a.crx, *cn_cry = a.cry, *cn_xfx = a.xfx, *cn_yfx = a.yfx;
  auto transport = [&](const PassRegion &rg, const PassRegion &rg_out, bool courant) -> int {
    if (rg.nk <= 0) return 0;
    // (a marching launch is as long as one wavefront's march whatever its size: below 16 levels -- the two sponge levels of the
    // reference defaults -- the LDS-tile kernel over the face is quicker)
    const bool march_flux = courant && rg.w == 0 && rg_out.w == 0 && fits && fused_ok && !a.use_cond && frame_fused_on() && flux_march_on() &&
                            g.geom != 2 && rg.nk >= 16;
    if (courant && !march_flux) RT(launch_pass(c, "dswc_d2", g.isd, g.ied, g.jsd, g.jed, PassRegion{0, rg.klist, rg.nk}, DswCubedD2{s}));
    // hybrid frame: delp, w, pt in ONE LDS-tile launch; the whole-face levels too unless a deln_flux damping has to get between
    // the transports (it changes the mass fluxes the later fields are weighted with)
    // the damped whole-face levels too: the del-2n fluxes of delp are formed first (they depend on delp alone) and the fused kernel
    // adds them to delp's fluxes before those weight w and pt; the damping of pt is added to its fluxes afterwards, that of w goes
    // to D4 as its own fluxes -- the order of the pass path below
    const bool full_ok = rg.w == 0 && rg_out.w == 0;
    if (march_flux) {
      // the damped whole-face levels on the marching kernel: Courant numbers, the three transports with delp's damping fluxes added
      // to its mass fluxes, the FLUXES as the result (DswTransportFused<..., FLUXES>); the frame along the edges by the frame kernel,
      // which overwrites what the march left there; the fields by D4, as after the passes
      const bool dv4 = c->lev_has_damp_v4;
      if (dv4) RT(deln(a.delp, nullptr, nullptr, nullptr, a.lv.nord_v, a.lv.damp_vt, 1.E-4, 0, c->lev_max_nord_v, nullptr, nullptr, rg));
      const double *dfx = dv4 ? cs_scratch(c, 5) : nullptr, *dfy = dv4 ? cs_scratch(c, 6) : nullptr;
      DswArgs am = a;
      am.uc = s.ut; am.vc = s.vt; am.mask_w = wo;
      am.dfx = dfx; am.dfy = dfy; am.dcoef = a.lv.damp_vt;
      am.ofx = s.fx; am.ofy = s.fy; am.ogxw = s.gxw; am.ogyw = s.gyw; am.ogx = s.gx; am.ogy = s.gy;
      MarchDims mf = make_march_dims(g, seg_rows(c, c->march_tj_fused, g.npz));
      mf.klist = rg.klist;
      const int nwf = mf.nwaves(rg.nk);
      RT(dispatch_hord(a.hord_dp, [&](auto H) {
        constexpr int HORD = decltype(H)::value;
        if (a.hydrostatic) return launch_w(c, "dswc_tp", nwf, DswTransportFused<HORD, false, true, 0, true>{g, am, mf});
        return launch_w(c, "dswc_tp", nwf, DswTransportFused<HORD, true, true, 0, true>{g, am, mf});
      }));
      TpfField fl[3];
      int nf = 0;
      fl[nf++] = TpfField{a.delp, s.fx, s.fy, a.hord_dp};
      if (!a.hydrostatic) fl[nf++] = TpfField{a.w, s.gxw, s.gyw, a.hord_vt};
      fl[nf++] = TpfField{a.pt, s.gx, s.gy, a.hord_tm};
      RT(tp2d_frame_fused(c, fl, nf, cn_crx, cn_cry, cn_xfx, cn_yfx, wo + 1, rg.klist, rg.nk, "dswc_tp", false, nullptr, nullptr, dfx, dfy,
                          dv4 ? a.lv.damp_vt : nullptr));
      if (c->lev_has_damp_t)    // :1014-1016: mass-weighted deln_flux inside fv_tp_2d(pt)
        RT(deln(a.pt, a.delp, s.gx, s.gy, a.lv.nord_t, a.lv.damp_t, 1.E-4, 0, c->lev_max_nord_t, nullptr, nullptr, rg));
      if (!a.hydrostatic && c->lev_has_w_damp_hi)   // :950-982: del6_vt_flux(w); nord_w = 0 is formed inside D4
        RT(deln(a.w, nullptr, nullptr, nullptr, a.lv.nord_w, a.lv.damp_w, 1.E-5, 1, c->lev_max_nord_w, s.wfx2, s.wfy2, rg));
      DswCubedState so = s;
      so.own_w = 0;
      RT(launch_pass(c, "dswc_d4", g.is, g.ie + 1, g.js, g.je + 1, rg_out, DswCubedD4{so}));
      return 0;
    }
    if (((rg.w > 0 && rg_out.w > 0) || full_ok) && !a.use_cond && frame_fused_on()) {
      TpfField fl[3];
      int nf = 0;
      fl[nf++] = TpfField{a.delp, s.fx, s.fy, a.hord_dp};
      if (!a.hydrostatic) fl[nf++] = TpfField{a.w, s.gxw, s.gyw, a.hord_vt};
      fl[nf++] = TpfField{a.pt, s.gx, s.gy, a.hord_tm};
      const bool dv4 = rg.w == 0 && c->lev_has_damp_v4;
      if (dv4)   // :919-920 without the final addition: raw fluxes in scratch 5, 6
        RT(deln(a.delp, nullptr, nullptr, nullptr, a.lv.nord_v, a.lv.damp_vt, 1.E-4, 0, c->lev_max_nord_v, nullptr, nullptr, rg));
      RT(tp2d_frame_fused(c, fl, nf, cn_crx, cn_cry, cn_xfx, cn_yfx, rg_out.w + 1, rg.klist, rg.nk, "dswc_tp", rg.w == 0, nullptr, nullptr,
                          dv4 ? cs_scratch(c, 5) : nullptr, dv4 ? cs_scratch(c, 6) : nullptr, dv4 ? a.lv.damp_vt : nullptr));
      if (rg.w == 0 && c->lev_has_damp_t)    // :1014-1016: mass-weighted deln_flux inside fv_tp_2d(pt)
        RT(deln(a.pt, a.delp, s.gx, s.gy, a.lv.nord_t, a.lv.damp_t, 1.E-4, 0, c->lev_max_nord_t, nullptr, nullptr, rg));
      if (rg.w == 0 && !a.hydrostatic && c->lev_has_w_damp_hi)   // :950-982: del6_vt_flux(w); nord_w = 0 is formed inside D4
        RT(deln(a.w, nullptr, nullptr, nullptr, a.lv.nord_w, a.lv.damp_w, 1.E-5, 1, c->lev_max_nord_w, s.wfx2, s.wfy2, rg));
      DswCubedState so = s;
      so.own_w = rg_out.w;
      RT(launch_pass(c, "dswc_d4", g.is, g.ie + 1, g.js, g.je + 1, rg_out, DswCubedD4{so}));
      return 0;
    }
    RT(tp2d_cubed(c, npz, a.delp, cn_crx, cn_cry, a.hord_dp, s.fx, s.fy, cn_xfx, cn_yfx, nullptr, nullptr, nullptr, nullptr, "dswc_tp", &rg));
    if (rg.w == 0 && c->lev_has_damp_v4)   // :919-920: deln_flux inside fv_tp_2d(delp) -- the mass fluxes carry it from here on
      RT(deln(a.delp, nullptr, s.fx, s.fy, a.lv.nord_v, a.lv.damp_vt, 1.E-4, 0, c->lev_max_nord_v, nullptr, nullptr, rg));
    if (!a.hydrostatic)
      RT(tp2d_cubed(c, npz, a.w, cn_crx, cn_cry, a.hord_vt, s.gxw, s.gyw, cn_xfx, cn_yfx, nullptr, nullptr, s.fx, s.fy, "dswc_tp", &rg));
    if (a.use_cond) {                      // :992-995: q_con with hord_dp, delp's mass fluxes and pt's damping
      RT(tp2d_cubed(c, npz, a.q_con, cn_crx, cn_cry, a.hord_dp, s.gxq, s.gyq, cn_xfx, cn_yfx, nullptr, nullptr, s.fx, s.fy, "dswc_tp", &rg));
      if (c->lev_has_damp_t)
        RT(deln(a.q_con, a.delp, s.gxq, s.gyq, a.lv.nord_t, a.lv.damp_t, 1.E-4, 0, c->lev_max_nord_t, nullptr, nullptr, rg));
    }
    RT(tp2d_cubed(c, npz, a.pt, cn_crx, cn_cry, a.hord_tm, s.gx, s.gy, cn_xfx, cn_yfx, nullptr, nullptr, s.fx, s.fy, "dswc_tp", &rg));
    if (rg.w == 0 && c->lev_has_damp_t)    // :1014-1016: mass-weighted deln_flux inside fv_tp_2d(pt)
      RT(deln(a.pt, a.delp, s.gx, s.gy, a.lv.nord_t, a.lv.damp_t, 1.E-4, 0, c->lev_max_nord_t, nullptr, nullptr, rg));
    if (rg.w == 0 && !a.hydrostatic && c->lev_has_w_damp_hi)   // :950-982: del6_vt_flux(w); nord_w = 0 is formed inside D4
      RT(deln(a.w, nullptr, nullptr, nullptr, a.lv.nord_w, a.lv.damp_w, 1.E-5, 1, c->lev_max_nord_w, s.wfx2, s.wfy2, rg));
    DswCubedState so = s;
    so.own_w = rg_out.w;
    RT(launch_pass(c, "dswc_d4", g.is, g.ie + 1, g.js, g.je + 1, rg_out, DswCubedD4{so}));
    return 0;
  };
  // part 0: the whole half; 1: what needs no Courant numbers (kinetic energy, vorticity, divergence damping: through D8); 2: the rest
  // (the vorticity transport, the wind update, heating)
  auto momentum = [&](const PassRegion &rg, const PassRegion &rg_out, int part) -> int {
    if (rg.nk <= 0) return 0;
    DswCubedState so = s;
    so.own_w = rg_out.w;
    if (part != 2) {
    RT(launch_pass(c, "dswc_ke", g.is, g.ie + 1, g.js, g.je + 1, rg, DswCubedD5{s}));
    RT(launch_pass(c, "dswc_d6", g.isd, g.ied + 1, g.jsd, g.jed + 1, rg, DswCubedD6{s}));
    // whole-face levels: the del-2n loop in one LDS-tile launch away from the face corners (cubed_dsw.h DswDampFused); the passes
    // keep the corner squares of 5 points (what fill_corners and the corner terms reach) and the rim their intermediates need
    const int wo_v = 5, wm_v = wo_v + DswDampFused::kMaxN + 1;
    const bool fused_v = rg.w == 0 && deln_fused_on() && c->lev_max_nord > 0 && c->lev_max_nord <= DswDampFused::kMaxN && npx == npy &&
                         npx - 1 >= 2 * wm_v + 8;
    auto vpass = [&](int i0, int i1, int j0, int j1, auto f) -> int {
      if (!fused_v) return launch_pass(c, L, i0, i1, j0, j1, rg, f);
      Dim3 gr;
      gr.x = 4; gr.y = 1; gr.z = (unsigned)rg.nk;
      return launch_p(c, L, gr, 0, CornerPass<decltype(f)>{i0, i1, j0, j1, wm_v, npx, npy, rg.klist, f});
    };
    for (int n = 1; n <= c->lev_max_nord; n++) {
      const bool may_fill = c->lev_max_nord - n != 0;   // some level may have nt /= 0 in this iteration
      const PassRegion rc{0, rg.klist, rg.nk};          // the corner fills: tiny boxes
      if (may_fill) RT(launch_pass(c, L, 1, 3, 1, 3, rc, DswCubedFillB{s, 1, n}));
      RT(vpass(g.is - 3, g.ie + 3, g.js - 3, g.je + 4, DswCubedDampVC{s, n, 0}));
      if (may_fill) RT(launch_pass(c, L, 1, 3, 1, 3, rc, DswCubedFillB{s, 2, n}));
      RT(vpass(g.is - 3, g.ie + 4, g.js - 3, g.je + 3, DswCubedDampVC{s, n, 1}));
      if (may_fill) {
        RT(launch_pass(c, L, 1, 3, 1, 3, rc, DswCubedFillD{s, 0, n}));
        RT(launch_pass(c, L, 1, 3, 1, 3, rc, DswCubedFillD{s, 1, n}));
      }
      RT(vpass(g.is - 2, g.ie + 3, g.js - 2, g.je + 3, DswCubedDampDiv{s, n}));
    }
    if (fused_v) {
      Dim3 gr;
      gr.x = (unsigned)((g.nx + 1 + DswDampFused::TI - 1) / DswDampFused::TI);
      gr.y = (unsigned)((g.ny + 1 + DswDampFused::TJ - 1) / DswDampFused::TJ);
      gr.z = (unsigned)rg.nk;
      RT(launch_p(c, L, gr, DswDampFused::lds_doubles, DswDampFused{s, wo_v, rg.klist}));
    }
    if (s.smag) {  // a2b_ord4 of the relative vorticity for the Smagorinsky coefficient (:1431-1440); scratch 0, 1 (c_sw's)
      A2bCubedState t;
      t.g = g; t.cg = c->cg; t.nf = 1; t.override_mask = 0;
      for (int f = 0; f < 4; f++) { t.in[f] = nullptr; t.out[f] = nullptr; t.qx[f] = t.qy[f] = nullptr; t.nlev[f] = 0; t.scale[f] = 1.0; t.top[f] = 0.; }
      t.in[0] = s.wk; t.out[0] = s.smag; t.nlev[0] = npz;
      if (!(t.qx[0] = cs_scratch(c, 0)) || !(t.qy[0] = cs_scratch(c, 1))) return fail("d_sw: out of device memory");
      if (rg.w == 0 && rg.nk == npz) {   // every level: the hybrid of nh_p_grad's a2b_ord4 (LDS-tile kernel + the passes on a frame)
        A2BCorners<32, 16> kc;
        kc.g = g;
        for (int f = 0; f < 4; f++) { kc.in[f] = nullptr; kc.out[f] = nullptr; kc.nlev[f] = 0; kc.scale[f] = 1.0; kc.top[f] = 0.; }
        kc.in[0] = s.wk; kc.out[0] = s.smag; kc.nlev[0] = npz;
        kc.nf = 1; kc.override_mask = 0;
        RT((run_a2b<32, 16>(c, kc, npz, "dswc_smag")));   // inside d_sw: the reference's D_SW timer, not PG_D
      } else {
        const PassRegion r{0, rg.klist, rg.nk};
        RT(launch_pass(c, "dswc_smag", 1, npx, 1, npy, r, A2bCubedPa{t}));
        RT(launch_pass(c, "dswc_smag", 1, npx, 1, npy, r, A2bCubedPb{t}));
      }
    }
    RT(launch_pass(c, L, g.is, g.ie + 1, g.js, g.je + 1, rg, DswCubedD7{so}));
    if (rg.w == 0 && c->lev_has_damp_v5)   // :1513-1515: del6_vt_flux of the RELATIVE vorticity (before D8 adds f0)
      // (work array 33, not the transports' 4: this chain may run beside theirs, see the lanes below)
      RT(deln(s.wk, nullptr, nullptr, nullptr, a.lv.nord_v, a.lv.damp_vt, 1.E-5, 1, c->lev_max_nord_v, s.dfx2, s.dfy2, rg, 33));
    RT(launch_pass(c, "dswc_d8", g.isd, g.ied, g.jsd, g.jed, rg, DswCubedD8{s}));
    }
    if (part == 1) return 0;
    if (frame_fused_on() && rg.w == 0 && fits && c->use_march && flux_march_on() && rg.nk >= 16) {
      // whole-face levels: the marching fv_tp_2d over the face, then the frame along the edges by the frame kernel (the fluxes are
      // not an input of either: the frame kernel simply overwrites what the march left there)
      MarchDims md = make_march_dims(g, seg_rows(c, c->march_tj, g.npz));
      md.klist = rg.klist;
      const int nwv = md.nwaves(rg.nk);
      RT(dispatch_hord(a.hord_vt, [&](auto H) {
        FluxMarch<decltype(H)::value> kf{g, md, s.wk, cn_crx, cn_cry, cn_xfx, cn_yfx, s.gx, s.gy};
        return launch_w(c, "dswc_tpv", nwv, kf);
      }));
      const TpfField fl[3] = {TpfField{s.wk, s.gx, s.gy, a.hord_vt}, TpfField{}, TpfField{}};
      RT(tp2d_frame_fused(c, fl, 1, cn_crx, cn_cry, cn_xfx, cn_yfx, wo + 1, rg.klist, rg.nk, "dswc_tpv", false));
    } else if (frame_fused_on()) {
      const TpfField fl[3] = {TpfField{s.wk, s.gx, s.gy, a.hord_vt}, TpfField{}, TpfField{}};
      RT(tp2d_frame_fused(c, fl, 1, cn_crx, cn_cry, cn_xfx, cn_yfx, rg_out.w + 1, rg.klist, rg.nk, "dswc_tpv", rg.w == 0));
    } else {
      RT(tp2d_cubed(c, npz, s.wk, cn_crx, cn_cry, a.hord_vt, s.gx, s.gy, cn_xfx, cn_yfx, nullptr, nullptr, nullptr, nullptr, "dswc_tpv", &rg));
    }
    RT(launch_pass(c, "dswc_d9", g.is, g.ie + 1, g.js, g.je + 1, rg_out, DswCubedD9{so}));
    if (rg.w == 0 && heat_pass) RT(launch_pass(c, "dswc_heat", g.is, g.ie, g.js, g.je, rg_out, DswCubedD10{so}));
    if (rg.w == 0 && c->lev_has_damp_v5) RT(launch_pass(c, "dswc_d9", g.is, g.ie + 1, g.js, g.je + 1, rg_out, DswCubedD11{so}));
    return 0;
  };

  // Two lanes (round 4): the marching kernels of the face interior are two long launches at one wavefront per SIMD; the ~35 pass / frame
  // launches (9 % of the points, 40 % of the time when they ran behind them) go to the side stream and run BESIDE them.  What orders
  // the two lanes: the passes of the levels the marching kernels do not take (the sponge levels: Courant numbers of their own, D2)
  // and the part of the momentum frame that needs no Courant numbers only read what D1 left -> after the fork; the frame of the
  // transports and the vorticity transport of the momentum frame read the Courant numbers the marching transport kernel writes for
  // its levels -> after `mid`; every output point and every accumulator is owned by exactly one kernel (mask_w / own_w, disjoint
  // levels), the passes' scratch arrays are not touched by the marching kernels.  Needs the same plain / damped level sets in both
  // halves (side_ok); not while profiling (the per-launch events would overlap).  FV3_MI355X_SIDE_STREAM=0: one lane.
  const bool lanes = hyb_t && hyb_m && c->side_ok && lanes_pay(c) && c->n_plain == c->n_plain_m;
  if (lanes) {
    RT(lane_prepare(c));
    const PassRegion fr_in{wm, c->klist, c->n_plain}, fr_out{wo, c->klist, c->n_plain};
    const PassRegion rest{0, c->klist + c->n_plain, c->n_damp};
    const PassRegion frm_in{wm, c->klist_m, c->n_plain_m}, frm_out{wo, c->klist_m, c->n_plain_m};
    const PassRegion restm{0, c->klist_m + c->n_plain_m, c->n_rest_m};
    DswArgs am = a;
    am.uc = s.ut; am.vc = s.vt; am.mask_w = wo;
    // FV3_MI355X_LANE_D2=1: the frame gets Courant numbers of its own (D2 on the frame, into scratch 29 .. 32, no accumulation: cx,
    // cy are the marching kernel's), so that EVERY pass runs beside the marching transport kernel (306 registers a wavefront: the
    // passes' wavefronts fit beside it; the marching momentum kernel fills the register file).  Default 0: the frame's transports
    // wait for the marching transport kernel's Courant numbers (`mid`) and run beside the momentum kernel.  Measured equal (3.99 -
    // 4.11 against 4.01 - 4.12 ms per C384 L127 pair, profiles/r04_v2_lanes_check.txt), so the variant without the extra pass and
    // the four extra work arrays is the default.
    static const int lane_d2 = [] { const char *e = std::getenv("FV3_MI355X_LANE_D2"); return e ? std::atoi(e) : 0; }();
    RT(lane_op(c, kLaneFork));
    RT(dsw_transport_march(c, am));                       // main: levels klist[0 : n_plain), Courant numbers too, whole face
    c->lane = 1;
    int rc = transport(rest, rest, true);
    if (!rc) rc = momentum(restm, restm, 0);
    if (!rc) rc = momentum(frm_in, frm_out, 1);
    if (!rc && lane_d2) {
      DswCubedState sf = s;
      double *cn[4];
      for (int n = 0; n < 4; n++)
        if (!(cn[n] = cs_scratch(c, 29 + n))) rc = fail("d_sw: out of device memory");
      if (!rc) {
        sf.a.crx = cn[0]; sf.a.cry = cn[1]; sf.a.xfx = cn[2]; sf.a.yfx = cn[3];
        DswCubedD2 d2{sf};
        d2.acc = 0;
        // (the frame's passes read Courant numbers up to one face beyond the frame of their intermediates)
        rc = launch_pass(c, "dswc_d2", g.isd, g.ied, g.jsd, g.jed, PassRegion{wm + 2, c->klist, c->n_plain}, d2);
        cn_crx = cn[0]; cn_cry = cn[1]; cn_xfx = cn[2]; cn_yfx = cn[3];
        if (!rc) rc = transport(fr_in, fr_out, false);
        if (!rc) rc = momentum(frm_in, frm_out, 2);
        cn_crx = a.crx; cn_cry = a.cry; cn_xfx = a.xfx; cn_yfx = a.yfx;
      }
    }
    c->lane = 0;
    if (rc) { (void)lane_op(c, kLaneJoin); return rc; }   // (a failed launch: still order the side stream's work before what follows)
    DswArgs mm = a;
    mm.mask_w = wo; mm.rsina = c->cg.rsina;
    if (lane_d2) {
      RT(lane_op(c, kLaneJoin));                          // (the frame's outputs and the marching momentum kernel's are disjoint,
      RT(dsw_momentum_march(c, mm));                      //  but nothing runs beside that kernel anyway: join first)
      return 0;
    }
    RT(lane_op(c, kLaneMid));
    RT(dsw_momentum_march(c, mm));                        // main: levels klist_m[0 : n_plain_m)
    c->lane = 1;
    rc = transport(fr_in, fr_out, false);
    if (!rc) rc = momentum(frm_in, frm_out, 2);
    c->lane = 0;
    const int rj = lane_op(c, kLaneJoin);
    if (rc) return rc;
    RT(rj);
    return 0;
  }
  // Every level damped (a production namelist: nord = 3, vtdm4, d_con, dddmp -- no level for the hybrid above): the two halves of d_sw
  // are chains of whole-face launches, and the momentum half up to the absolute vorticity (kinetic energy, vorticity, the del-2n loop of
  // the divergence, Smagorinsky, the vorticity's del6_vt_flux) reads nothing the transport half writes -- it runs beside it on the side
  // lane; what needs the Courant numbers (the vorticity transport, the wind update, heating) follows on the main lane after the join.
  // The two chains share no work array (the vorticity's damping chain has its own, 33).
  if (!hyb_t && !hyb_m && !a.use_cond && frame_fused_on() && lanes_pay(c)) {   // (the pass forms of fv_tp_2d share work arrays 4 .. 7)
    RT(lane_prepare(c));
    const PassRegion all{0, nullptr, npz};
    RT(lane_op(c, kLaneFork));
    c->lane = 1;
    int rc = momentum(all, all, 1);
    c->lane = 0;
    if (!rc) rc = transport(all, all, true);
    const int rj = lane_op(c, kLaneJoin);
    if (rc) return rc;
    RT(rj);
    // (the rest of the momentum half on the side lane too -- behind an event after the transports' Courant numbers, with flux arrays of
    // its own -- was measured no faster, and it is not independent: D4 leaves the heat of the w damping in heat_source, which the
    // heating pass adds to)
    RT(momentum(all, all, 2));
    return 0;
  }
  if (hyb_t) {
    DswArgs am = a;
    am.uc = s.ut; am.vc = s.vt; am.mask_w = wo;
    RT(dsw_transport_march(c, am));                       // levels klist[0 : n_plain): Courant numbers too, whole face
    RT(transport(PassRegion{wm, c->klist, c->n_plain}, PassRegion{wo, c->klist, c->n_plain}, false));
    RT(transport(PassRegion{0, c->klist + c->n_plain, c->n_damp}, PassRegion{0, c->klist + c->n_plain, c->n_damp}, true));
  } else {
    RT(transport(PassRegion{0, nullptr, npz}, PassRegion{0, nullptr, npz}, true));
  }
  if (hyb_m) {
    DswArgs am = a;
    am.mask_w = wo; am.rsina = c->cg.rsina;
    RT(dsw_momentum_march(c, am));                        // levels klist_m[0 : n_plain_m)
    RT(momentum(PassRegion{wm, c->klist_m, c->n_plain_m}, PassRegion{wo, c->klist_m, c->n_plain_m}, 0));
    RT(momentum(PassRegion{0, c->klist_m + c->n_plain_m, c->n_rest_m}, PassRegion{0, c->klist_m + c->n_plain_m, c->n_rest_m}, 0));
  } else {
    RT(momentum(PassRegion{0, nullptr, npz}, PassRegion{0, nullptr, npz}, 0));
  }
  return 0;
}

// phase 0: the whole routine; 1: only the part that needs no halo of uc, vc, divg_d (interior strips / segments of the
// fused transport kernel); 2: everything else, after phase 1
static int d_sw_impl(fv3_ctx *c, const fv3_dsw_params *p, double *delpc, const double *delp, const double *pt,
                     const double *u, const double *v, const double *w, const double *uc, const double *vc,
                     const double *ua, const double *va, const double *divg_d, double *mfx, double *mfy,
                     double *cx, double *cy, double *crx, double *cry, double *xfx, double *yfx,
                     const double *q_con, double *delp_out, double *pt_out, double *u_out, double *v_out,
                     double *w_out, double *q_con_out, double *heat_s, double *diss_e, int phase) {
  if (!c || !c->grid_ready) return fail("fv3_d_sw: context has no grid (call fv3_grid_upload)");
  if (!c->lev_ready) return fail("fv3_d_sw: per-level coefficients missing (call fv3_dsw_levels_upload)");
  if (!p) return fail("fv3_d_sw: null params");
  if (!tp_ord_supported(p->hord_dp) || !tp_ord_supported(p->hord_vt) || !tp_ord_supported(p->hord_tm))
    return fail("fv3_d_sw: hord_dp/vt/tm must be one of 5,-5,6,8,10");
  if (!sw_ord_supported(p->hord_mt)) return fail("fv3_d_sw: hord_mt must be in 5..11");
  if (!p->hydrostatic && (!w || !w_out)) return fail("fv3_d_sw: nonhydrostatic call needs w and w_out");
  if (p->use_cond && (!q_con || !q_con_out)) return fail("fv3_d_sw: use_cond needs q_con and q_con_out");
  if (delp == delp_out || pt == pt_out || u == u_out || v == v_out || (w && w == w_out))
    return fail("fv3_d_sw: *_out buffers must not alias the inputs");
  if ((!heat_s && c->lev_has_dcon) || (!diss_e && c->g.do_diss_est))
    return fail("fv3_d_sw: heat_s = NULL with d_con > 1e-5 on some level / diss_e = NULL with do_diss_est: the caller reads them (dyn_core.F90:798-812)");
  const Grid &g = c->g;
  const int npz = g.npz;
  DswArgs a;
  a.dt = p->dt;
  a.hord_tr = p->hord_tr; a.hord_mt = p->hord_mt; a.hord_vt = p->hord_vt; a.hord_tm = p->hord_tm; a.hord_dp = p->hord_dp;
  a.dddmp = p->dddmp; a.d4_bg = p->d4_bg; a.kgb = p->kgb;
  a.hydrostatic = p->hydrostatic; a.use_cond = p->use_cond;
  a.lv = DswLevels{c->lev_i, c->lev_i + npz, c->lev_i + 2 * npz, c->lev_i + 3 * npz,
                   c->lev_d, c->lev_d + npz, c->lev_d + 2 * npz, c->lev_d + 3 * npz, c->lev_d + 4 * npz};
  a.delp = delp; a.pt = pt; a.u = u; a.v = v; a.w = w; a.uc = uc; a.vc = vc; a.ua = ua; a.va = va;
  a.divg_d = divg_d; a.q_con = q_con;
  a.mfx = mfx; a.mfy = mfy; a.cx = cx; a.cy = cy; a.crx = crx; a.cry = cry; a.xfx = xfx; a.yfx = yfx;
  a.delp_out = delp_out; a.pt_out = pt_out; a.u_out = u_out; a.v_out = v_out; a.w_out = w_out;
  a.q_con_out = q_con_out; a.heat_s = heat_s; a.diss_e = diss_e; a.delpc = delpc;

  if (is_cubed(c)) {
    lev_activate(c, 0);
    if (phase == 1) return 0;  // no interior / rest split on a face: everything runs in 'rest' (or the unsplit call)
    if (!a.heat_s || !a.diss_e) {
      for (int n = 0; n < 2; n++)
        if (!c->heat_scr[n]) RT(rt_malloc((void **)&c->heat_scr[n], sizeof(double) * g.nCC() * npz));
      a.skip_heat = (!a.heat_s && !a.diss_e) ? 1 : 0;   // the marching kernel of the face interior then stores neither
      if (!a.heat_s) a.heat_s = c->heat_scr[0];
      if (!a.diss_e) a.diss_e = c->heat_scr[1];
    }
    return dsw_cubed(c, a);
  }
  // the fused marching kernel forms the Courant numbers itself for its levels
  const bool fused = c->use_march && c->use_fused && !a.use_cond && a.hord_dp == a.hord_tm &&
                     (a.hydrostatic || a.hord_dp == a.hord_vt);
  constexpr int TI = FV3_DSW_TI, TJ = FV3_DSW_TJ;
  const bool march = c->use_march != 0;
  // marching momentum: no Smagorinsky coefficient, no dissipation estimate (level conditions in klist_m)
  const bool march_m = march && a.dddmp < 1.E-5 && !g.do_diss_est;
  // the sponge levels on the marching kernels: their branch-free forms with uniform metrics, both halves of d_sw marching
  lev_activate(c, (FV3_BF && c->sponge_march && g.geom == 2 && fused && march_m) ? 1 : 0);
  // heat_s, diss_e = NULL: the caller does not read them (dyn_core.F90:798-812 reads them when d_con > 1e-5 or do_diss_est; in the
  // reference they are 2-D work arrays private to a level).  The branch-free marching kernels then do not store them; every other kernel
  // writes (and the damped levels' momentum kernels read) real arrays, the context's own
  const bool all_bf = FV3_BF && fused && march_m && c->n_damp == 0 && c->n_rest_m == 0;
  if ((!a.heat_s || !a.diss_e) && !all_bf) {
    for (int n = 0; n < 2; n++)
      if (!c->heat_scr[n]) RT(rt_malloc((void **)&c->heat_scr[n], sizeof(double) * g.nCC() * npz));
    if (!a.heat_s) a.heat_s = c->heat_scr[0];
    if (!a.diss_e) a.diss_e = c->heat_scr[1];
  }

  auto courant = [&]() -> int {  // Courant numbers and area fluxes of the levels the fused kernel does not take
    if (fused && c->n_damp == 0) return 0;
    DswCourant kf{g, a, fused ? c->klist + c->n_plain : nullptr};
    const size_t nmax = g.nCX() > g.nCY() ? g.nCX() : g.nCY();
    Dim3 grid;
    grid.x = (unsigned)((nmax + DswCourant::CH - 1) / DswCourant::CH);
    grid.y = 1;
    grid.z = (unsigned)(fused ? c->n_damp : npz);
    return launch_p(c, "d_sw_courant", grid, 0, kf);
  };
  auto tile_transport = [&]() -> int {
    if (march && c->n_damp == 0) return 0;
    DswTransport<TI, TJ> kf{g, a, march ? c->klist + c->n_plain : nullptr};
    Dim3 grid;
    DswTransport<TI, TJ>::grid_dims(g, grid.x, grid.y);
    grid.z = (unsigned)(march ? c->n_damp : npz);
    return launch_p(c, "d_sw_transport", grid, DswTransport<TI, TJ>::lds_doubles, kf);
  };
  auto tile_momentum = [&]() -> int {
    if (march_m && c->n_rest_m == 0) return 0;
    DswMomentum<TI, TJ> kf{g, a, march_m ? c->klist_m + c->n_plain_m : nullptr};
    Dim3 grid;
    DswMomentum<TI, TJ>::grid_dims(g, grid.x, grid.y);
    grid.z = (unsigned)(march_m ? c->n_rest_m : npz);
    return launch_p(c, "d_sw_momentum", grid, DswMomentum<TI, TJ>::lds_doubles, kf);
  };

  // Side stream: when the tile kernels only take a few levels (the sponge layers) and both halves of d_sw route the
  // same levels there, their chain courant -> transport -> momentum touches data disjoint from the marching kernels'
  // and runs concurrently with them.  (Not while profiling: the per-kernel events live on the main stream.)
  // region of the fused transport kernel this call runs (phases 1/2 split it only when it is the active path)
  const bool split = fused && march && c->n_plain > 0 && dsw_has_interior(c);
  if (phase == 1) {
    if (split) RT(dsw_transport_march(c, a, 1));
    return 0;
  }
  const int region = (phase == 2 && split) ? 2 : 0;
  const bool side = fused && march_m && c->side_ok && c->use_side && !c->prof_on && !c->grp && c->n_damp > 0 && c->n_plain > 0;
  if (side) {
    if (!c->stream2) {
      RT(rt_stream_create(&c->stream2));
      RT(rt_event_create(&c->ev_fork));
      RT(rt_event_create(&c->ev_join));
    }
    const stream_t main_stream = c->stream;
    RT(rtf_event_record(c->ev_fork, main_stream));
    RT(rtf_stream_wait_event(c->stream2, c->ev_fork));
    c->stream = c->stream2;
    int rc = courant();
    if (!rc) rc = tile_transport();
    if (!rc) rc = tile_momentum();
    c->stream = main_stream;
    if (int rc2 = rtf_event_record(c->ev_join, c->stream2)) return rc2;
    RT(rc);
    RT(dsw_transport_march(c, a, region));
    RT(dsw_momentum_march(c, a));
    RT(rtf_stream_wait_event(main_stream, c->ev_join));
    return 0;
  }
  RT(courant());
  if (march && c->n_plain > 0) RT(dsw_transport_march(c, a, region));
  RT(tile_transport());
  if (march_m && c->n_plain_m > 0) RT(dsw_momentum_march(c, a));
  RT(tile_momentum());
  return 0;
}

#define FV3_DSW_ARGS                                                                                                      \
  c, p, delpc, delp, pt, u, v, w, uc, vc, ua, va, divg_d, mfx, mfy, cx, cy, crx, cry, xfx, yfx, q_con, delp_out, pt_out, \
      u_out, v_out, w_out, q_con_out, heat_s, diss_e
#define FV3_DSW_PARAMS                                                                                                  \
  fv3_ctx *c, const fv3_dsw_params *p, double *delpc, const double *delp, const double *pt, const double *u,           \
      const double *v, const double *w, const double *uc, const double *vc, const double *ua, const double *va,        \
      const double *divg_d, double *mfx, double *mfy, double *cx, double *cy, double *crx, double *cry, double *xfx,   \
      double *yfx, const double *q_con, double *delp_out, double *pt_out, double *u_out, double *v_out, double *w_out, \
      double *q_con_out, double *heat_s, double *diss_e
extern "C" int fv3_d_sw(FV3_DSW_PARAMS) { return d_sw_impl(FV3_DSW_ARGS, 0); }
extern "C" int fv3_d_sw_interior(FV3_DSW_PARAMS) { return d_sw_impl(FV3_DSW_ARGS, 1); }
extern "C" int fv3_d_sw_rest(FV3_DSW_PARAMS) { return d_sw_impl(FV3_DSW_ARGS, 2); }

// ---- periodic halo fill (single rank owns the whole doubly periodic tile) ------------------------
struct HaloPeriodic {
  Grid g;
  double *f;
  int kind;  // 0=A 1=U 2=V 3=B
  static constexpr int CH = 1024;
  FV3_HD void operator()(int bx, int, int bz, int tid, double *) const {
    const int ni = g.nid + ((kind == 2 || kind == 3) ? 1 : 0), nj = g.njd + ((kind == 1 || kind == 3) ? 1 : 0);
    double *s = f + (size_t)bz * ni * nj;
    for (int idx = bx * CH + tid; idx < (bx + 1) * CH; idx += kNT) {
      if (idx >= ni * nj) continue;
      const int i = g.isd + idx % ni, j = g.jsd + idx / ni;
      // the staggered edge row/column (je+1 / ie+1) belongs to the compute domain of a
      // north-east staggered field and is not a halo point
      const int ihi = g.ie + ((kind == 2 || kind == 3) ? 1 : 0), jhi = g.je + ((kind == 1 || kind == 3) ? 1 : 0);
      if (i >= g.is && i <= ihi && j >= g.js && j <= jhi) continue;
      int si = i, sj = j;
      if (si < g.is) si += g.nx; else if (si > ihi) si -= g.nx;
      if (sj < g.js) sj += g.ny; else if (sj > jhi) sj -= g.ny;
      s[idx] = s[(size_t)(sj - g.jsd) * ni + (si - g.isd)];
    }
  }
};

extern "C" int fv3_halo_fill_periodic(fv3_ctx *c, double *field, int kind, int nk) {
  if (!c) return fail("fv3_halo_fill_periodic: null ctx");
  if (kind < 0 || kind > 3) return fail("fv3_halo_fill_periodic: bad kind");
  const Grid &g = c->g;
  if (g.nx < NG + 1 || g.ny < NG + 1) return fail("fv3_halo_fill_periodic: tile smaller than the halo");
  HaloPeriodic kf{g, field, kind};
  const size_t n = (size_t)(g.nid + 1) * (g.njd + 1);
  Dim3 grid;
  grid.x = (unsigned)((n + HaloPeriodic::CH - 1) / HaloPeriodic::CH);
  grid.y = 1;
  grid.z = (unsigned)nk;
  RT(launch_p(c, "halo_periodic", grid, 0, kf));
  return 0;
}

// ---- multi-rank halo exchange: pack / unpack ---------------------------------------------------------
struct HaloStrip {  // one (message, field) pair
  double *field;
  long buf_off;             // element offset of this field's strip inside the message buffer
  int i0, ni, j0, nj;       // strip origin (array indices) and extent
  int pitch, slab, nk, dir; // array leading dimension, k-slab size, levels, message index
  int si0, sj0;             // HaloSelf: origin of the strip this one is filled from (same field, same extent)
};
struct HaloCopy {
  HaloStrip st[8 * FV3_HALO_MAX_FIELDS];
  double *buf[8];
  int pack;  // 1: field -> buffer, 0: buffer -> field
  static constexpr int CH = 2048;
  FV3_HD void operator()(int bx, int, int bz, int tid, double *) const {
    const HaloStrip &s = st[bz];
    const long n = (long)s.ni * s.nj * s.nk;
    double *b = buf[s.dir] + s.buf_off;
    for (long idx = (long)bx * CH + tid; idx < (long)(bx + 1) * CH && idx < n; idx += kNT) {
      const int i = (int)(idx % s.ni), j = (int)((idx / s.ni) % s.nj), k = (int)(idx / ((long)s.ni * s.nj));
      double *f = s.field + (size_t)k * s.slab + (size_t)(s.j0 + j) * s.pitch + (s.i0 + i);
      if (pack) b[idx] = *f; else *f = b[idx];
    }
  }
};

// One rank, doubly periodic: every message is a message to myself, so the halo strip of side -d is filled straight from the send
// strip of side d -- what pack + unpack do through a buffer, in one launch for the whole group
struct HaloSelf {
  HaloStrip st[8 * FV3_HALO_MAX_FIELDS];
  static constexpr int CH = 2048;
  FV3_HD void operator()(int bx, int, int bz, int tid, double *) const {
    const HaloStrip &s = st[bz];
    const long n = (long)s.ni * s.nj * s.nk;
    for (long idx = (long)bx * CH + tid; idx < (long)(bx + 1) * CH && idx < n; idx += kNT) {
      const int i = (int)(idx % s.ni), j = (int)((idx / s.ni) % s.nj), k = (int)(idx / ((long)s.ni * s.nj));
      s.field[(size_t)k * s.slab + (size_t)(s.j0 + j) * s.pitch + (s.i0 + i)] =
          s.field[(size_t)k * s.slab + (size_t)(s.sj0 + j) * s.pitch + (s.si0 + i)];
    }
  }
};

// index ranges along one direction (Fortran indices), off = -1, 0, +1; s = stagger of that direction
static void halo_range(int lo, int hi, int s, int off, bool send, int &a, int &b) {
  if (off == 0) { a = lo; b = hi + s; return; }
  if (send) {
    if (off < 0) { a = lo + s; b = lo + s + NG - 1; } else { a = hi - NG + 1; b = hi; }
  } else {
    if (off < 0) { a = lo - NG; b = lo - 1; } else { a = hi + s + 1; b = hi + s + NG; }
  }
}

static int halo_build(fv3_ctx *c, int nfields, const fv3_halo_field *fields, bool pack, HaloCopy &hc, size_t elems[8],
                      long &maxn) {
  if (!c || !c->grid_ready) return fail("fv3_halo: context has no grid");
  if (nfields < 1 || nfields > FV3_HALO_MAX_FIELDS) return fail("fv3_halo: 1..%d fields per group", FV3_HALO_MAX_FIELDS);
  const Grid &g = c->g;
  if (g.nx < NG + 1 || g.ny < NG + 1) return fail("fv3_halo: block smaller than the halo");
  maxn = 0;
  int n = 0, d = 0;
  for (int dj = -1; dj <= 1; dj++)
    for (int di = -1; di <= 1; di++) {
      if (di == 0 && dj == 0) continue;
      long off = 0;
      for (int f = 0; f < nfields; f++) {
        const int kind = fields[f].kind;
        if (kind < 0 || kind > 3 || !fields[f].field) return fail("fv3_halo: bad field %d", f);
        const int si = (kind == 2 || kind == 3) ? 1 : 0, sj = (kind == 1 || kind == 3) ? 1 : 0;
        int ia, ib, ja, jb;
        // pack: the send strip of side d; unpack: message d fills the halo of side -d
        halo_range(g.is, g.ie, si, pack ? di : -di, pack, ia, ib);
        halo_range(g.js, g.je, sj, pack ? dj : -dj, pack, ja, jb);
        HaloStrip &s = hc.st[n++];
        s.field = fields[f].field;
        s.buf_off = off;
        s.i0 = ia - g.isd; s.ni = ib - ia + 1;
        s.j0 = ja - g.jsd; s.nj = jb - ja + 1;
        s.pitch = g.nid + si;
        s.slab = (g.nid + si) * (g.njd + sj);
        s.nk = fields[f].nk;
        s.dir = d;
        const long cnt = (long)s.ni * s.nj * s.nk;
        if (cnt > maxn) maxn = cnt;
        off += cnt;
      }
      elems[d++] = (size_t)off;
    }
  hc.pack = pack ? 1 : 0;
  return 0;
}

extern "C" int fv3_halo_message_elems(fv3_ctx *c, int nfields, const fv3_halo_field *fields, size_t elems[8]) {
  HaloCopy hc;
  long maxn;
  return halo_build(c, nfields, fields, true, hc, elems, maxn);
}

static int halo_copy(fv3_ctx *c, int nfields, const fv3_halo_field *fields, double *const buf[8], bool pack) {
  HaloCopy hc;
  size_t elems[8];
  long maxn;
  if (halo_build(c, nfields, fields, pack, hc, elems, maxn)) return 1;
  for (int d = 0; d < 8; d++) {
    if (!buf[d]) return fail("fv3_halo: null message buffer %d", d);
    hc.buf[d] = buf[d];
  }
  Dim3 grid;
  grid.x = (unsigned)((maxn + HaloCopy::CH - 1) / HaloCopy::CH);
  grid.y = 1;
  grid.z = (unsigned)(8 * nfields);
  RT(launch_p(c, pack ? "halo_pack" : "halo_unpack", grid, 0, hc));
  return 0;
}
extern "C" int fv3_halo_periodic_group(fv3_ctx *c, int nfields, const fv3_halo_field *fields) {
  HaloCopy src, dst;
  size_t elems[8];
  long maxn;
  if (halo_build(c, nfields, fields, true, src, elems, maxn)) return 1;
  if (halo_build(c, nfields, fields, false, dst, elems, maxn)) return 1;
  HaloSelf hs;
  for (int n = 0; n < 8 * nfields; n++) {   // strip n of both lists: message d, field f
    hs.st[n] = dst.st[n];
    hs.st[n].si0 = src.st[n].i0;
    hs.st[n].sj0 = src.st[n].j0;
  }
  Dim3 grid;
  grid.x = (unsigned)((maxn + HaloSelf::CH - 1) / HaloSelf::CH);
  grid.y = 1;
  grid.z = (unsigned)(8 * nfields);
  RT(launch_p(c, "halo_periodic", grid, 0, hs));
  return 0;
}
extern "C" int fv3_halo_pack(fv3_ctx *c, int nfields, const fv3_halo_field *fields, double *const sendbuf[8]) {
  return halo_copy(c, nfields, fields, sendbuf, true);
}
extern "C" int fv3_halo_unpack(fv3_ctx *c, int nfields, const fv3_halo_field *fields, const double *const recvbuf[8]) {
  return halo_copy(c, nfields, fields, const_cast<double *const *>(recvbuf), false);
}

// ---- peer exchange behind the C ABI: RCCL send / recv on a stream the context owns ------------------------------------------
// What start_group_halo_update / complete_group_halo_update (tools/fv_mp_mod.F90:646-876) and mp_reduce_max (:1683) do over
// FMS / MPI.  librccl is loaded at run time (dlopen) the first time a communicator is made, so the library has no link
// dependency on it and a host process that already carries an RCCL (PyTorch) keeps using that one.
#ifndef FV3_HOST_EMU
#include <dlfcn.h>
namespace {
struct Id128 { char b[128]; };  // ncclUniqueId, passed by value
struct RcclApi {
  void *lib = nullptr;
  int (*GetUniqueId)(void *) = nullptr;
  int (*CommInitRank)(void **, int, Id128, int) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;
int rccl_load() {
  if (g_rccl.lib) return 0;
  void *h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return fail("fv3_comm: cannot load librccl.so (%s)", dlerror());
  auto sym = [&](const char *n) { return dlsym(h, n); };
  *(void **)&g_rccl.GetUniqueId = sym("ncclGetUniqueId");
  *(void **)&g_rccl.CommInitRank = sym("ncclCommInitRank");
  *(void **)&g_rccl.CommDestroy = sym("ncclCommDestroy");
  *(void **)&g_rccl.GroupStart = sym("ncclGroupStart");
  *(void **)&g_rccl.GroupEnd = sym("ncclGroupEnd");
  *(void **)&g_rccl.Send = sym("ncclSend");
  *(void **)&g_rccl.Recv = sym("ncclRecv");
  *(void **)&g_rccl.AllReduce = sym("ncclAllReduce");
  *(void **)&g_rccl.GetErrorString = sym("ncclGetErrorString");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.GroupStart || !g_rccl.GroupEnd || !g_rccl.Send || !g_rccl.Recv ||
      !g_rccl.AllReduce)
    return fail("fv3_comm: librccl.so lacks an entry point");
  g_rccl.lib = h;
  return 0;
}
constexpr int kNcclDouble = 8, kNcclMax = 2;  // ncclFloat64, ncclMax (rccl.h)
constexpr int kNcclInt64 = 4, kNcclSum = 0;   // ncclInt64, ncclSum
#define NC(x)                                                                                              \
  do {                                                                                                     \
    int e_ = (x);                                                                                          \
    if (e_) return fail("RCCL: %s (%s)", g_rccl.GetErrorString ? g_rccl.GetErrorString(e_) : "error", #x); \
  } while (0)
}  // namespace
#endif

#ifdef FV3_HOST_EMU
// ---- the message transport of the logic harness (tests/hostemu ONLY; the product is the RCCL code above) --------------------------
// What the exchange does around a message -- pack lists, the order of the sends and receives of a group, the unpack -- is the same
// code in both builds; only the wire differs.  Here a message is a file in a directory named by the "unique id": the k-th message
// rank a sends to rank b is <dir>/m_<a>_<b>_<k>, written under another name and renamed (so a reader never sees half of it); the
// k-th receive rank b posts for a waits for exactly that file and REFUSES a message of another size.  That is the matching rule of
// ncclSend / ncclRecv inside a group (per pair of ranks, in posting order), so several processes on the CPU exercise what a
// loopback on one GPU cannot: a sender and a receiver that disagree about the order or the content of their messages.
#include <sys/stat.h>
#include <unistd.h>
namespace {
struct EmuComm {
  std::string dir;
  int rank, n;
  std::vector<long> sseq, rseq;
  long coll;
};
int emu_wait_file(const std::string &name) {
  struct stat sb;
  for (long spin = 0; spin < 600000; spin++) {   // up to ~2 minutes
    if (stat(name.c_str(), &sb) == 0) return 0;
    usleep(200);
  }
  return fail("host-emulation transport: %s never arrived (the peer posts its messages in another order, or died)", name.c_str());
}
int emu_put(const std::string &name, const void *buf, size_t bytes) {
  const std::string tmp = name + ".tmp";
  FILE *f = std::fopen(tmp.c_str(), "wb");
  if (!f) return fail("host-emulation transport: cannot write %s", tmp.c_str());
  const size_t w = bytes ? std::fwrite(buf, 1, bytes, f) : 0;
  std::fclose(f);
  if (w != bytes || std::rename(tmp.c_str(), name.c_str())) return fail("host-emulation transport: write of %s failed", name.c_str());
  return 0;
}
int emu_get(const std::string &name, void *buf, size_t bytes, bool remove) {
  if (emu_wait_file(name)) return 1;
  struct stat sb;
  if (stat(name.c_str(), &sb) || (size_t)sb.st_size != bytes)
    return fail("host-emulation transport: %s carries %ld bytes, the receive expects %ld (the two ends of a link disagree)", name.c_str(),
                (long)sb.st_size, (long)bytes);
  FILE *f = std::fopen(name.c_str(), "rb");
  if (!f) return fail("host-emulation transport: cannot read %s", name.c_str());
  const size_t r = bytes ? std::fread(buf, 1, bytes, f) : 0;
  std::fclose(f);
  if (r != bytes) return fail("host-emulation transport: short read of %s", name.c_str());
  if (remove) std::remove(name.c_str());
  return 0;
}
int emu_send(EmuComm *e, const void *buf, size_t bytes, int peer) {
  char nm[64];
  std::snprintf(nm, sizeof nm, "/m_%d_%d_%ld", e->rank, peer, e->sseq[peer]++);
  return emu_put(e->dir + nm, buf, bytes);
}
int emu_recv(EmuComm *e, void *buf, size_t bytes, int peer) {
  char nm[64];
  std::snprintf(nm, sizeof nm, "/m_%d_%d_%ld", peer, e->rank, e->rseq[peer]++);
  return emu_get(e->dir + nm, buf, bytes, true);
}
// every rank's `bytes` of `in`, rank by rank, into out (n * bytes)
int emu_allgather(EmuComm *e, const void *in, size_t bytes, std::vector<char> &out) {
  char nm[64];
  const long k = e->coll++;
  std::snprintf(nm, sizeof nm, "/c_%ld_%d", k, e->rank);
  if (emu_put(e->dir + nm, in, bytes)) return 1;
  out.resize((size_t)e->n * bytes);
  for (int r = 0; r < e->n; r++) {
    std::snprintf(nm, sizeof nm, "/c_%ld_%d", k, r);
    if (emu_get(e->dir + nm, out.data() + (size_t)r * bytes, bytes, false)) return 1;
  }
  return 0;
}
}  // namespace
#endif

extern "C" int fv3_comm_get_unique_id(unsigned char *id) {
  if (!id) return fail("fv3_comm_get_unique_id: null");
#ifdef FV3_HOST_EMU
  std::memset(id, 0, FV3_COMM_ID_BYTES);
  char tmpl[] = "/tmp/fv3emu_XXXXXX";
  if (!mkdtemp(tmpl)) return fail("fv3_comm_get_unique_id: mkdtemp failed");
  std::memcpy(id, tmpl, sizeof tmpl);
  return 0;
#else
  if (rccl_load()) return 1;
  NC(g_rccl.GetUniqueId(id));
  return 0;
#endif
}

extern "C" int fv3_comm_init(fv3_ctx *c, int rank, int nranks, const unsigned char *id) {
  if (!c || !id || nranks < 1 || rank < 0 || rank >= nranks) return fail("fv3_comm_init: bad argument");
  if (c->comm) return fail("fv3_comm_init: the context already has a communicator");
  c->comm_rank = rank;
  c->comm_size = nranks;
  RT(rt_stream_create(&c->comm_stream));
  RT(rt_event_create(&c->ev_packed));
  RT(rt_event_create(&c->ev_arrived));
#ifdef FV3_HOST_EMU
  {
    EmuComm *e = new (std::nothrow) EmuComm();
    if (!e) return fail("fv3_comm_init: out of host memory");
    e->dir = id[0] ? std::string(reinterpret_cast<const char *>(id), strnlen(reinterpret_cast<const char *>(id), FV3_COMM_ID_BYTES)) : std::string();
    e->rank = rank; e->n = nranks; e->coll = 0;
    e->sseq.assign(nranks, 0); e->rseq.assign(nranks, 0);
    // (an all-zero id has no directory: the messages would go to the filesystem root -- for one rank as well)
    if (e->dir.empty()) { delete e; return fail("fv3_comm_init: the id of fv3_comm_get_unique_id is needed (host emulation: its directory)"); }
    c->comm = (void *)e;
  }
#else
  if (rccl_load()) return 1;
  Id128 uid;
  std::memcpy(uid.b, id, 128);
  NC(g_rccl.CommInitRank(&c->comm, nranks, uid, rank));
#endif
  return 0;
}

extern "C" int fv3_comm_destroy(fv3_ctx *c) {
  if (!c) return 0;
#ifndef FV3_HOST_EMU
  if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
#else
  // (the exchange directory /tmp/fv3emu_* stays: other ranks / contexts of the communicator may still read from it -- tests only)
  if (c->comm) delete static_cast<EmuComm *>(c->comm);
#endif
  c->comm = nullptr;
  for (int d = 0; d < 8; d++) {
    if (c->msg_send[d]) rt_free(c->msg_send[d]);
    if (c->msg_recv[d]) rt_free(c->msg_recv[d]);
    c->msg_send[d] = c->msg_recv[d] = nullptr;
    c->msg_cap[d] = 0;
  }
  if (c->ev_packed) rt_event_destroy(c->ev_packed);
  if (c->ev_arrived) rt_event_destroy(c->ev_arrived);
  if (c->comm_stream) rt_stream_destroy(c->comm_stream);
  c->ev_packed = c->ev_arrived = nullptr;
  c->comm_stream = nullptr;
  return 0;
}

// start_group_halo_update: pack every field of the group (one kernel), then -- on the communication stream, which waits for
// the pack kernel only -- one grouped send + receive per neighbour offset d (message d goes to the rank at offset d and fills
// its (-d)-side halo; to[d] / from[d] are the ranks at offsets d / -d, fv3_halo_message_elems order).  Kernels launched
// between start and complete on the context's stream overlap the transfers.
extern "C" int fv3_halo_start(fv3_ctx *c, int nfields, const fv3_halo_field *fields, const int *to, const int *from) {
  if (!c || !c->comm) return fail("fv3_halo_start: call fv3_comm_init first");
  if (!fields || !to || !from || nfields < 1 || nfields > FV3_HALO_MAX_FIELDS) return fail("fv3_halo_start: bad argument");
  if (c->pend_n) return fail("fv3_halo_start: the previous group has not been completed");
  for (int d = 0; d < 8; d++) {   // checked BEFORE the pack and the group: a bad entry must not leave an RCCL group open
    if (to[d] < 0 || to[d] >= c->comm_size || from[d] < 0 || from[d] >= c->comm_size) return fail("fv3_halo_start: peer out of range");
  }
  size_t elems[8];
  if (fv3_halo_message_elems(c, nfields, fields, elems)) return 1;
  for (int d = 0; d < 8; d++) {
    if (elems[d] > c->msg_cap[d]) {
      if (c->msg_send[d]) rt_free(c->msg_send[d]);
      if (c->msg_recv[d]) rt_free(c->msg_recv[d]);
      RT(rt_malloc((void **)&c->msg_send[d], sizeof(double) * elems[d]));
      RT(rt_malloc((void **)&c->msg_recv[d], sizeof(double) * elems[d]));
      c->msg_cap[d] = elems[d];
    }
  }
  if (fv3_halo_pack(c, nfields, fields, c->msg_send)) return 1;
  RT(rtf_event_record(c->ev_packed, c->stream));
  RT(rtf_stream_wait_event(c->comm_stream, c->ev_packed));
#ifdef FV3_HOST_EMU
  {   // the group: every send, then every receive, each matched per peer in posting order (see the transport above)
    EmuComm *e = static_cast<EmuComm *>(c->comm);
    for (int d = 0; d < 8; d++)
      if (emu_send(e, c->msg_send[d], sizeof(double) * elems[d], to[d])) return 1;
    for (int d = 0; d < 8; d++)
      if (emu_recv(e, c->msg_recv[d], sizeof(double) * elems[d], from[d])) return 1;
  }
#else
  NC(g_rccl.GroupStart());
  int rc = 0;
  for (int d = 0; d < 8 && !rc; d++) {
    rc = g_rccl.Send(c->msg_send[d], elems[d], kNcclDouble, to[d], c->comm, c->comm_stream);
    if (!rc) rc = g_rccl.Recv(c->msg_recv[d], elems[d], kNcclDouble, from[d], c->comm, c->comm_stream);
  }
  const int rc2 = g_rccl.GroupEnd();   // the group is closed on every path
  if (rc || rc2) return fail("RCCL: %s (halo group)", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc ? rc : rc2) : "error");
#endif
  RT(rtf_event_record(c->ev_arrived, c->comm_stream));
  for (int f = 0; f < nfields; f++) c->pend_fields[f] = fields[f];
  c->pend_n = nfields;
  return 0;
}

// complete_group_halo_update: the context's stream waits for the transfers and unpacks them into the halos
extern "C" int fv3_halo_complete(fv3_ctx *c) {
  if (!c || !c->pend_n) return fail("fv3_halo_complete: no group in flight");
  RT(rtf_stream_wait_event(c->stream, c->ev_arrived));
  const int n = c->pend_n;
  c->pend_n = 0;
  return fv3_halo_unpack(c, n, c->pend_fields, c->msg_recv);
}

// ---- the cube-edge exchange: one face per rank (or several faces per rank) -------------------------------------------------------
// mpp_update_domains on the six-tile mosaic (tools/fv_mp_mod.F90:498-546; group updates :646-876) and mpp_get_boundary of (u, v)
// (model/dyn_core.F90:1151-1163) as peer messages: per pair of faces ONE message per group of fields; sender and receiver walk the
// RECEIVING face's table (cube_topo.h) in the same order, so a message needs no header; the sign of the component rotation is applied
// while packing.
struct CubePlanDev {
  int n_send, n_recv;
  int *s_member, *s_sign, *s_seg, *r_member, *r_seg;   // device
  long *s_idx, *r_idx;                                 // device
  int send_cnt[6], recv_cnt[6], send_start[6], recv_start[6];
};
static void cube_plans_free(fv3_ctx *c) {
  for (int n = 0; n < 5; n++) {
    CubePlanDev *p = c->cube_plan[n];
    if (!p) continue;
    int *ip[5] = {p->s_member, p->s_sign, p->s_seg, p->r_member, p->r_seg};
    for (int *q : ip) if (q) rt_free(q);
    if (p->s_idx) rt_free(p->s_idx);
    if (p->r_idx) rt_free(p->r_idx);
    delete p;
    c->cube_plan[n] = nullptr;
  }
  if (c->cube_send) rt_free(c->cube_send);
  if (c->cube_recv) rt_free(c->cube_recv);
  c->cube_send = c->cube_recv = nullptr;
  c->cube_cap_send = c->cube_cap_recv = 0;
}
extern "C" long fv3_cube_table(int npx, int ng, int kind, int member, int face, long *dst, int *src_face, int *comp, long *src, int *sign) {
  if (npx < 3 || ng < 1 || kind < 0 || kind > 4 || face < 0 || face > 5 || member < 0 || member >= CubeTopo::members(kind)) return -1;
  const std::vector<CubeRow> rows = CubeTopo(npx, ng).table(kind, member, face);
  if (dst)
    for (size_t r = 0; r < rows.size(); r++) {
      dst[r] = rows[r].dst; src_face[r] = rows[r].tile; comp[r] = rows[r].comp; src[r] = rows[r].src; sign[r] = rows[r].sign;
    }
  return (long)rows.size();
}
static int cube_plan_get(fv3_ctx *c, int face, int kind, CubePlanDev **out) {
  if (c->cube_face >= 0 && c->cube_face != face) return fail("fv3_cube_halo: the context was planned as face %d", c->cube_face);
  if (c->cube_face == face && c->cube_plan[kind]) { *out = c->cube_plan[kind]; return 0; }
  const CubeTopo topo(c->g.npx, NG);
  const int nm = CubeTopo::members(kind);
  std::vector<int> sm, ss, sseg, rm, rseg;
  std::vector<long> si, ri;
  CubePlanDev *p = new (std::nothrow) CubePlanDev();
  if (!p) return fail("fv3_cube_halo: out of host memory");
  // what this face owes face r: the rows of r's table with tile == face, r ascending, members, table order
  for (int r = 0; r < 6; r++) {
    p->send_start[r] = (int)si.size();
    if (r != face)
      for (int m = 0; m < nm; m++)
        for (const CubeRow &row : topo.table(kind, m, r))
          if (row.tile == face) { sm.push_back(row.comp ? 1 - m : m); si.push_back(row.src); ss.push_back(row.sign); sseg.push_back(r); }
    p->send_cnt[r] = (int)si.size() - p->send_start[r];
  }
  // what this face receives from face s: the rows of its own table with tile == s, in the same order
  std::vector<std::vector<CubeRow>> mine(nm);
  for (int m = 0; m < nm; m++) mine[m] = topo.table(kind, m, face);
  for (int sf = 0; sf < 6; sf++) {
    p->recv_start[sf] = (int)ri.size();
    for (int m = 0; m < nm; m++)
      for (const CubeRow &row : mine[m])
        if (row.tile == sf) { rm.push_back(m); ri.push_back(row.dst); rseg.push_back(sf); }
    p->recv_cnt[sf] = (int)ri.size() - p->recv_start[sf];
  }
  p->n_send = (int)si.size();
  p->n_recv = (int)ri.size();
  auto up_i = [&](int **d, const std::vector<int> &h) -> int {
    if (h.empty()) { *d = nullptr; return 0; }
    if (rt_malloc((void **)d, sizeof(int) * h.size())) return 1;
    return rtf_h2d(*d, h.data(), sizeof(int) * h.size(), c->stream);
  };
  auto up_l = [&](long **d, const std::vector<long> &h) -> int {
    if (h.empty()) { *d = nullptr; return 0; }
    if (rt_malloc((void **)d, sizeof(long) * h.size())) return 1;
    return rtf_h2d(*d, h.data(), sizeof(long) * h.size(), c->stream);
  };
  // built into a plan of its own: a failure frees what was uploaded and leaves the context as it was (not pinned to `face`)
  p->s_member = p->s_sign = p->s_seg = p->r_member = p->r_seg = nullptr;
  p->s_idx = p->r_idx = nullptr;
  int rc_up = up_i(&p->s_member, sm) || up_i(&p->s_sign, ss) || up_i(&p->s_seg, sseg) || up_l(&p->s_idx, si) ||
              up_i(&p->r_member, rm) || up_i(&p->r_seg, rseg) || up_l(&p->r_idx, ri);
  if (!rc_up) rc_up = rtf_sync(c->stream);
  if (rc_up) {
    int *ip[5] = {p->s_member, p->s_sign, p->s_seg, p->r_member, p->r_seg};
    for (int *q : ip) if (q) rt_free(q);
    if (p->s_idx) rt_free(p->s_idx);
    if (p->r_idx) rt_free(p->r_idx);
    delete p;
    return fail("fv3_cube_halo: the exchange plan could not be uploaded (out of device memory?)");
  }
  c->cube_face = face;
  c->cube_plan[kind] = p;
  *out = p;
  return 0;
}
static void cube_planes(const Grid &g, int kind, size_t pl[2]) {
  if (kind == kCubeA) pl[0] = pl[1] = g.nA();
  else if (kind == kCubeB) pl[0] = pl[1] = g.nB();
  else if (kind == kCubeC) { pl[0] = g.nV(); pl[1] = g.nU(); }
  else { pl[0] = g.nU(); pl[1] = g.nV(); }
}
struct CubePack {   // thread per (row, level): message layout [peer][field][level][row of the peer's segment]
  int n, nk, vector;
  const int *member, *sign, *seg;
  const long *idx;
  const double *f0, *f1;
  size_t pl0, pl1;
  int cnt[6], start[6];
  size_t off[6];
  double *buf;
  static constexpr int CH = 1024;
  FV3_HD void operator()(int bx, int, int, int tid, double *) const {
    const long tot = (long)n * nk;
    for (long e = (long)bx * CH + tid; e < (long)(bx + 1) * CH && e < tot; e += kNT) {
      const int row = (int)(e % n), k = (int)(e / n), sg = seg[row];
      double v = member[row] ? f1[(size_t)k * pl1 + idx[row]] : f0[(size_t)k * pl0 + idx[row]];
      if (vector && sign[row] < 0) v = -v;
      buf[off[sg] + (size_t)k * cnt[sg] + (row - start[sg])] = v;
    }
  }
};
struct CubeUnpack {
  int n, nk;
  const int *member, *seg;
  const long *idx;
  double *f0, *f1;
  size_t pl0, pl1;
  int cnt[6], start[6];
  size_t off[6];
  const double *buf;
  static constexpr int CH = 1024;
  FV3_HD void operator()(int bx, int, int, int tid, double *) const {
    const long tot = (long)n * nk;
    for (long e = (long)bx * CH + tid; e < (long)(bx + 1) * CH && e < tot; e += kNT) {
      const int row = (int)(e % n), k = (int)(e / n), sg = seg[row];
      const double v = buf[off[sg] + (size_t)k * cnt[sg] + (row - start[sg])];
      if (member[row]) f1[(size_t)k * pl1 + idx[row]] = v; else f0[(size_t)k * pl0 + idx[row]] = v;
    }
  }
};
// nctx contexts = the faces this rank holds (faces[] ascending); ctxs[0] carries the communicator (fv3_comm_init); face_rank[t] = the
// rank holding face t; fields[i * nfields + f] = field f of context i (the same kinds / nk / flags on every context and rank).
extern "C" int fv3_cube_halo_start(int nctx, fv3_ctx *const *ctxs, const int *faces, const int *face_rank, int nfields,
                                   const fv3_cube_field *fields) {
  if (nctx < 1 || nctx > 6 || !ctxs || !faces || !face_rank || !fields || nfields < 1 || nfields > FV3_HALO_MAX_FIELDS)
    return fail("fv3_cube_halo_start: bad argument");
  fv3_ctx *c0 = ctxs[0];
  if (!c0 || !c0->comm) return fail("fv3_cube_halo_start: call fv3_comm_init on the first context");
  // every argument is checked before anything is packed or posted (a bad peer must not leave an RCCL group open)
  for (int i = 0; i < nctx; i++) {
    if (!ctxs[i] || !ctxs[i]->grid_ready || ctxs[i]->g.grid_type >= 3) return fail("fv3_cube_halo_start: context %d is not a cubed-sphere face", i);
    if (faces[i] < 0 || faces[i] > 5 || (i > 0 && faces[i] <= faces[i - 1])) return fail("fv3_cube_halo_start: faces must be ascending in 0..5");
    if (ctxs[i]->cube_pend_n) return fail("fv3_cube_halo_start: the previous group of face %d has not been completed", faces[i]);
    if (face_rank[faces[i]] != c0->comm_rank) return fail("fv3_cube_halo_start: face_rank does not place face %d on this rank", faces[i]);
    for (int f = 0; f < nfields; f++) {
      const fv3_cube_field &fd = fields[i * nfields + f];
      if (fd.kind < 0 || fd.kind > 4 || !fd.f0 || (fd.kind >= kCubeD && !fd.f1) || fd.nk < 1) return fail("fv3_cube_halo_start: bad field %d", f);
    }
  }
  for (int t = 0; t < 6; t++)
    if (face_rank[t] < 0 || face_rank[t] >= c0->comm_size) return fail("fv3_cube_halo_start: rank of face %d out of range", t);
  // ---- plans, buffer layout, pack ----
  size_t soff[6][FV3_HALO_MAX_FIELDS][6], scount[6][6], rcount[6][6];
  for (int i = 0; i < nctx; i++) {
    fv3_ctx *c = ctxs[i];
    CubePlanDev *pl[FV3_HALO_MAX_FIELDS];
    for (int f = 0; f < nfields; f++)
      if (cube_plan_get(c, faces[i], fields[i * nfields + f].kind, &pl[f])) return 1;
    size_t so = 0, ro = 0;
    for (int r = 0; r < 6; r++) {      // peer-major: one contiguous message per peer face
      scount[i][r] = rcount[i][r] = 0;
      for (int f = 0; f < nfields; f++) {
        const int nk = fields[i * nfields + f].nk;
        soff[i][f][r] = so;
        so += (size_t)pl[f]->send_cnt[r] * nk;
        scount[i][r] += (size_t)pl[f]->send_cnt[r] * nk;
        c->cube_roff[f][r] = ro;
        ro += (size_t)pl[f]->recv_cnt[r] * nk;
        rcount[i][r] += (size_t)pl[f]->recv_cnt[r] * nk;
      }
    }
    if (so > c->cube_cap_send) {
      if (c->cube_send) rt_free(c->cube_send);
      RT(rt_malloc((void **)&c->cube_send, sizeof(double) * so));
      c->cube_cap_send = so;
    }
    if (ro > c->cube_cap_recv) {
      if (c->cube_recv) rt_free(c->cube_recv);
      RT(rt_malloc((void **)&c->cube_recv, sizeof(double) * ro));
      c->cube_cap_recv = ro;
    }
    for (int f = 0; f < nfields; f++) {
      const fv3_cube_field &fd = fields[i * nfields + f];
      if (pl[f]->n_send == 0) continue;
      CubePack kf;
      kf.n = pl[f]->n_send; kf.nk = fd.nk; kf.vector = fd.scalar_pair ? 0 : 1;
      kf.member = pl[f]->s_member; kf.sign = pl[f]->s_sign; kf.seg = pl[f]->s_seg; kf.idx = pl[f]->s_idx;
      kf.f0 = fd.f0; kf.f1 = fd.f1 ? fd.f1 : fd.f0;
      size_t pls[2];
      cube_planes(c->g, fd.kind, pls);
      kf.pl0 = pls[0]; kf.pl1 = pls[1];
      for (int r = 0; r < 6; r++) { kf.cnt[r] = pl[f]->send_cnt[r]; kf.start[r] = pl[f]->send_start[r]; kf.off[r] = soff[i][f][r]; }
      kf.buf = c->cube_send;
      Dim3 grid;
      grid.x = (unsigned)(((long)kf.n * kf.nk + CubePack::CH - 1) / CubePack::CH); grid.y = 1; grid.z = 1;
      RT(launch_p(c, "cube_pack", grid, 0, kf));
    }
    if (!c->ev_packed) RT(rt_event_create(&c->ev_packed));
    RT(rtf_event_record(c->ev_packed, c->stream));
    RT(rtf_stream_wait_event(c0->comm_stream, c->ev_packed));
  }
  // ---- the messages: sends ordered by (sender face, receiver face), receives likewise -- the same order on both ends of a link ----
#ifdef FV3_HOST_EMU
  {   // the same order as the RCCL group below: sends by (sender face, receiver face), receives by (sender face, receiver face)
    EmuComm *e = static_cast<EmuComm *>(c0->comm);
    for (int i = 0; i < nctx; i++)
      for (int r = 0; r < 6; r++)
        if (scount[i][r] && emu_send(e, ctxs[i]->cube_send + soff[i][0][r], sizeof(double) * scount[i][r], face_rank[r])) return 1;
    for (int sf = 0; sf < 6; sf++)
      for (int i = 0; i < nctx; i++)
        if (rcount[i][sf] && emu_recv(e, ctxs[i]->cube_recv + ctxs[i]->cube_roff[0][sf], sizeof(double) * rcount[i][sf], face_rank[sf])) return 1;
  }
#else
  NC(g_rccl.GroupStart());
  int rc = 0;
  for (int i = 0; i < nctx && !rc; i++)
    for (int r = 0; r < 6 && !rc; r++)
      if (scount[i][r]) rc = g_rccl.Send(ctxs[i]->cube_send + soff[i][0][r], scount[i][r], kNcclDouble, face_rank[r], c0->comm, c0->comm_stream);
  for (int sf = 0; sf < 6 && !rc; sf++)
    for (int i = 0; i < nctx && !rc; i++)
      if (rcount[i][sf]) rc = g_rccl.Recv(ctxs[i]->cube_recv + ctxs[i]->cube_roff[0][sf], rcount[i][sf], kNcclDouble, face_rank[sf], c0->comm, c0->comm_stream);
  const int rc2 = g_rccl.GroupEnd();   // closed on every path
  if (rc || rc2) return fail("RCCL: %s (cube-edge group)", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc ? rc : rc2) : "error");
#endif
  RT(rtf_event_record(c0->ev_arrived, c0->comm_stream));
  for (int i = 0; i < nctx; i++) {
    for (int f = 0; f < nfields; f++) ctxs[i]->cube_pend[f] = fields[i * nfields + f];
    ctxs[i]->cube_pend_n = nfields;
  }
  return 0;
}
extern "C" int fv3_cube_halo_complete(int nctx, fv3_ctx *const *ctxs) {
  if (nctx < 1 || !ctxs || !ctxs[0]) return fail("fv3_cube_halo_complete: bad argument");
  fv3_ctx *c0 = ctxs[0];
  for (int i = 0; i < nctx; i++)
    if (!ctxs[i] || !ctxs[i]->cube_pend_n) return fail("fv3_cube_halo_complete: no group in flight");
  for (int i = 0; i < nctx; i++) {
    fv3_ctx *c = ctxs[i];
    RT(rtf_stream_wait_event(c->stream, c0->ev_arrived));
    const int nf = c->cube_pend_n;
    c->cube_pend_n = 0;
    for (int f = 0; f < nf; f++) {
      const fv3_cube_field &fd = c->cube_pend[f];
      CubePlanDev *pl = c->cube_plan[fd.kind];
      if (!pl || pl->n_recv == 0) continue;
      CubeUnpack kf;
      kf.n = pl->n_recv; kf.nk = fd.nk;
      kf.member = pl->r_member; kf.seg = pl->r_seg; kf.idx = pl->r_idx;
      kf.f0 = fd.f0; kf.f1 = fd.f1 ? fd.f1 : fd.f0;
      size_t pls[2];
      cube_planes(c->g, fd.kind, pls);
      kf.pl0 = pls[0]; kf.pl1 = pls[1];
      for (int r = 0; r < 6; r++) { kf.cnt[r] = pl->recv_cnt[r]; kf.start[r] = pl->recv_start[r]; kf.off[r] = c->cube_roff[f][r]; }
      kf.buf = c->cube_recv;
      Dim3 grid;
      grid.x = (unsigned)(((long)kf.n * kf.nk + CubeUnpack::CH - 1) / CubeUnpack::CH); grid.y = 1; grid.z = 1;
      RT(launch_p(c, "cube_unpack", grid, 0, kf));
    }
  }
  return 0;
}

// mp_reduce_max (tools/fv_mp_mod.F90:1683): element-wise maximum of n host doubles over the ranks, in place
extern "C" int fv3_allreduce_max(fv3_ctx *c, double *buf, int n) {
  if (!c || !c->comm || !buf || n < 1) return fail("fv3_allreduce_max: bad argument / no communicator");
  if (c->comm_size == 1) return 0;
#ifdef FV3_HOST_EMU
  {
    std::vector<char> all;
    if (emu_allgather(static_cast<EmuComm *>(c->comm), buf, sizeof(double) * n, all)) return 1;
    const double *a = reinterpret_cast<const double *>(all.data());
    for (int r = 0; r < c->comm_size; r++)
      for (int m = 0; m < n; m++) buf[m] = a[(size_t)r * n + m] > buf[m] ? a[(size_t)r * n + m] : buf[m];
    return 0;
  }
#else
  double *d = nullptr;
  RT(rt_malloc((void **)&d, sizeof(double) * n));
  RT(rtf_h2d(d, buf, sizeof(double) * n, c->comm_stream));
  NC(g_rccl.AllReduce(d, d, (size_t)n, kNcclDouble, kNcclMax, c->comm, c->comm_stream));
  RT(rtf_d2h(buf, d, sizeof(double) * n, c->comm_stream));
  RT(rtf_sync(c->comm_stream));
  rt_free(d);
  return 0;
#endif
}

// g_sum(..., reproduce = .true.) (model/fv_grid_utils.F90:2879-2925) = FMS's mpp_global_sum(flags = BITWISE_EFP_SUM): the
// extended-fixed-point sum (Hallberg & Adcroft 2014; FMS mpp/mpp_efp.F90, a dependency outside the reference tree): every addend
// as six integer digits of radix 2^46 (2^92 .. 2^-138), integer sums (exact: independent of order and of the rank layout),
// carries, back to a double from the most significant digit down.  values: n HOST doubles (the caller's p*area of its compute
// domain); with a communicator of several ranks the digits are all-reduced.  The same algorithm as global_sum.py.
extern "C" int fv3_ordered_sum(fv3_ctx *c, const double *values, size_t n, double *sum) {
  if (!values || !sum) return fail("fv3_ordered_sum: null argument");
  constexpr int NI = 6, NB = 46;
  const double R = (double)(1LL << NB);
  const double pr[NI] = {R * R, R, 1.0, 1.0 / R, 1.0 / (R * R), 1.0 / (R * R * R)};
  __int128 acc[NI] = {0, 0, 0, 0, 0, 0};
  for (size_t m = 0; m < n; m++) {
    const double a = values[m];
    if (!(a == a) || a > 1.7e308 || a < -1.7e308) return fail("fv3_ordered_sum: non-finite addend");
    double rs = a < 0. ? -a : a;
    // the range of the extended fixed point number: NUMBIT bits above the leading radix (FMS mpp_efp.F90 aborts with an overflow)
    if (rs >= pr[0] * R) return fail("fv3_ordered_sum: addend out of the range of the extended-fixed-point sum (|a| >= 2**138)");
    for (int i = 0; i < NI; i++) {
      const double iv = std::floor(rs * (1.0 / pr[i]));
      rs = rs - iv * pr[i];
      acc[i] += a < 0. ? -(__int128)iv : (__int128)iv;
    }
  }
  auto carry = [&](__int128 *d) {
    for (int i = NI - 1; i >= 1; i--) {
      __int128 cy = d[i] >> NB;   // arithmetic shift = floor division: the remainder is in [0, 2^46)
      d[i] -= cy << NB;
      d[i - 1] += cy;
    }
  };
  carry(acc);
  // the leading digit travels as int64 through the all-reduce and comes back as a double: |d0| < 2**40 leaves room for the sum
  // over the ranks (the guard of global_sum.py)
  const __int128 lim = (__int128)1 << 40;
  if (acc[0] >= lim || acc[0] <= -lim) return fail("fv3_ordered_sum: leading digit too large (|sum| >= 2**132)");
  long long dig[NI];
  for (int i = 0; i < NI; i++) dig[i] = (long long)acc[i];
  if (c && c->comm && c->comm_size > 1) {
#ifdef FV3_HOST_EMU
    {
      std::vector<char> all;
      if (emu_allgather(static_cast<EmuComm *>(c->comm), dig, sizeof(long long) * NI, all)) return 1;
      const long long *a = reinterpret_cast<const long long *>(all.data());
      for (int i = 0; i < NI; i++) {
        acc[i] = 0;
        for (int r = 0; r < c->comm_size; r++) acc[i] += a[(size_t)r * NI + i];
      }
      carry(acc);
      for (int i = 0; i < NI; i++) dig[i] = (long long)acc[i];
    }
#else
    long long *d = nullptr;
    RT(rt_malloc((void **)&d, sizeof(long long) * NI));
    RT(rtf_h2d(d, dig, sizeof(long long) * NI, c->comm_stream));
    NC(g_rccl.AllReduce(d, d, (size_t)NI, kNcclInt64, kNcclSum, c->comm, c->comm_stream));
    RT(rtf_d2h(dig, d, sizeof(long long) * NI, c->comm_stream));
    RT(rtf_sync(c->comm_stream));
    rt_free(d);
    for (int i = 0; i < NI; i++) acc[i] = dig[i];
    carry(acc);
    for (int i = 0; i < NI; i++) dig[i] = (long long)acc[i];
#endif
  }
  double r = 0.;
  for (int i = 0; i < NI; i++) r = r + pr[i] * (double)dig[i];
  *sum = r;
  return 0;
}

// prt_maxmin / prt_mxm (tools/fv_diagnostics.F90:4213-4313): out = {max * fac, min * fac, gmean * fac}; gmean = g_sum(q(:,:,nk), area,
// mode = 1) (fv_grid_utils.F90:2879-2925: the local sum in the reference's loop order over the global area, itself an EFP sum).  With
// a communicator of several ranks the extrema and the sums are reduced (mp_reduce_min / _max / _sum).
extern "C" int fv3_prt_maxmin(fv3_ctx *c, const double *q, int nk, double fac, double out[3]) {
  if (!c || !c->grid_ready || !q || nk < 1 || !out) return fail("fv3_prt_maxmin: bad context/arguments");
  const Grid &g = c->g;
  std::vector<double> mm(2 * nk), lev(g.nA());
  {
    double *d = nullptr;
    RT(rt_malloc((void **)&d, sizeof(double) * 2 * nk));
    struct Free { double *p; ~Free() { rt_free(p); } } free_d{d};   // on every way out
    LevelMinMax kf{g, q, d};
    RT(launch_p(c, "prt_maxmin", Dim3{1, 1, (unsigned)nk}, 2 * kNT, kf));
    RT(rtf_d2h(mm.data(), d, sizeof(double) * 2 * nk, c->stream));
    RT(rtf_d2h(lev.data(), q + (size_t)(nk - 1) * g.nA(), sizeof(double) * g.nA(), c->stream));
    if (c->host_area.empty()) {   // (forgotten by fv3_grid_upload / fv3_grid_upload_cubed: a new grid brings a new area)
      c->host_area.resize(g.nA());
      RT(rtf_d2h(c->host_area.data(), g.area, sizeof(double) * g.nA(), c->stream));
    }
    RT(rtf_sync(c->stream));
  }
  double qmin = mm[0], qmax = mm[1];
  for (int k = 1; k < nk; k++) {
    qmin = mm[2 * k] < qmin ? mm[2 * k] : qmin;
    qmax = mm[2 * k + 1] > qmax ? mm[2 * k + 1] : qmax;
  }
  // gmean as the reference's g_sum forms it: over the area the ranks of this context's communicator hold together.  A rank that holds
  // several faces as separate contexts WITHOUT a communicator (six contexts on one GPU) gets the mean over the one face of `c`: the
  // caller combines the faces (area-weighted; the faces of the gnomonic cube have equal areas).
  if (c->global_area == 0.) {   // g_sum's saved global_area: mpp_global_sum(area, BITWISE_EFP_SUM)
    std::vector<double> ar((size_t)g.nx * g.ny);
    for (int j = g.js; j <= g.je; j++)
      for (int i = g.is; i <= g.ie; i++) ar[(size_t)(j - g.js) * g.nx + (i - g.is)] = c->host_area[g.iA(i, j)];
    RT(fv3_ordered_sum(c, ar.data(), ar.size(), &c->global_area));
  }
  double gsum = 0.;
  for (int j = g.js; j <= g.je; j++)
    for (int i = g.is; i <= g.ie; i++) gsum = gsum + lev[g.iA(i, j)] * c->host_area[g.iA(i, j)];
  if (c->comm && c->comm_size > 1) {
    double ext[2] = {qmax, -qmin};
    RT(fv3_allreduce_max(c, ext, 2));
    qmax = ext[0]; qmin = -ext[1];
    RT(fv3_ordered_sum(c, &gsum, 1, &gsum));   // mp_reduce_sum of the ranks' partial sums (exact, so independent of their order)
  }
  out[0] = qmax * fac;
  out[1] = qmin * fac;
  out[2] = gsum / c->global_area * fac;
  return 0;
}

// ---- table-driven halo gather: the cubed-sphere face-to-face updates (index reversal, u <-> v with sign) ---------------------
// One entry = one halo value: ptrs[dst_sel][k * stride[dst_sel] + dst_idx] = sign * ptrs[src_sel][k * stride[src_sel] + src_idx].
// The tables come from the host (gfdl_atmos_cubed_sphere_amd/cubed_sphere.py: CubeTopology); with all six faces on one GPU one
// launch fills every halo of a field (pair); across GPUs the same tables drive the pack (dst = message buffer) and unpack sides.
struct fv3_gather {
  int n;
  int *tab;  // device: dst_sel, dst_idx, src_sel, src_idx, sign  (5 x n)
};
struct GatherKernel {
  int n;
  const int *tab;
  double *ptr[16];
  size_t stride[16];
  static constexpr int CH = 1024;
  FV3_HD void operator()(int bx, int, int bz, int tid, double *) const {
    const int k = bz;
    for (int e = bx * CH + tid; e < (bx + 1) * CH && e < n; e += kNT) {
      const int ds = tab[e], di = tab[n + e], ss = tab[2 * n + e], si = tab[3 * n + e], sg = tab[4 * n + e];
      const double v = ptr[ss][(size_t)k * stride[ss] + si];
      ptr[ds][(size_t)k * stride[ds] + di] = sg < 0 ? -v : v;
    }
  }
};
extern "C" int fv3_gather_create(fv3_ctx *c, int n, const int *dst_sel, const int *dst_idx, const int *src_sel,
                                 const int *src_idx, const int *sign, fv3_gather **out) {
  if (!c || !out || n <= 0 || !dst_sel || !dst_idx || !src_sel || !src_idx || !sign) return fail("fv3_gather_create: bad argument");
  for (int e = 0; e < n; e++)
    if (dst_sel[e] < 0 || dst_sel[e] > 15 || src_sel[e] < 0 || src_sel[e] > 15) return fail("fv3_gather_create: selector out of range");
  fv3_gather *t = new (std::nothrow) fv3_gather();
  if (!t) return fail("fv3_gather_create: out of host memory");
  t->n = n;
  t->tab = nullptr;
  if (rt_malloc((void **)&t->tab, sizeof(int) * 5 * (size_t)n)) { delete t; return fail("fv3_gather_create: out of device memory"); }
  const int *src[5] = {dst_sel, dst_idx, src_sel, src_idx, sign};
  for (int m = 0; m < 5; m++) RT(rtf_h2d(t->tab + (size_t)m * n, src[m], sizeof(int) * (size_t)n, c->stream));
  RT(rtf_sync(c->stream));
  *out = t;
  return 0;
}
extern "C" int fv3_gather_destroy(fv3_gather *t) {
  if (t) {
    if (t->tab) rt_free(t->tab);
    delete t;
  }
  return 0;
}
extern "C" int fv3_gather_run(fv3_ctx *c, const fv3_gather *t, int nk, int nptr, double *const *ptrs, const size_t *strides) {
  if (!c || !t || !ptrs || !strides || nptr < 1 || nptr > 16 || nk < 1) return fail("fv3_gather_run: bad argument");
  GatherKernel kf;
  kf.n = t->n;
  kf.tab = t->tab;
  for (int m = 0; m < 16; m++) {
    kf.ptr[m] = m < nptr ? ptrs[m] : nullptr;
    kf.stride[m] = m < nptr ? strides[m] : 0;
  }
  Dim3 grid;
  grid.x = (unsigned)((t->n + GatherKernel::CH - 1) / GatherKernel::CH);
  grid.y = 1;
  grid.z = (unsigned)nk;
  // one launch that reads and writes every face: whatever the faces of a group have queued goes first, and it is not queued itself
  fv3_group *grp = c->grp;
  if (grp) RT(grp_pump(grp, true));
  c->grp = nullptr;
  const int rc_g = launch_p(c, "halo_gather", grid, 0, kf);
  c->grp = grp;
  RT(rc_g);
  return 0;
}

// ================================================================================================
// nonhydrostatic column path
// ================================================================================================
static int need_scratch(fv3_ctx *c, int n) {
  // A x (km+1), and room for the (nx+2) x (ny+2) columns of the C-grid solver in blocks of 64 (scr_col layout)
  const size_t nblk = (((size_t)(c->g.nx + 2) * (c->g.ny + 2) + 63) / 64) * 64;
  const size_t bytes = (c->g.nA() > nblk ? c->g.nA() : nblk) * (size_t)(c->g.npz + 1) * sizeof(double);
  for (int s = 0; s < n; s++)
    if (!c->scratch[s]) RT(rt_malloc((void **)&c->scratch[s], bytes));
  return 0;
}

// pooled column launch (FV3_COL_FOR_POOL): pool workgroups, or one per column block when there are fewer blocks / no pool
static int col_pool(const fv3_ctx *c, int ncol) {
  const int nb = (ncol + 255) / 256;
  return (c->col_pool > 0 && c->riem_blocked && nb > c->col_pool) ? c->col_pool : 0;
}
static Dim3 col_grid(int ncol) {
  Dim3 gr;
  gr.x = (unsigned)((ncol + 255) / 256);
  gr.y = 1;
  gr.z = 1;
  return gr;
}

extern "C" int fv3_set_dp_ref(fv3_ctx *c, const double *dp0) {
  if (!c || !dp0) return fail("fv3_set_dp_ref: null argument");
  const int km = c->g.npz;
  if (km < 2) return fail("fv3_set_dp_ref: needs npz >= 2");
  if (!c->dp0) RT(rt_malloc((void **)&c->dp0, sizeof(double) * km));
  const size_t n_edge = (size_t)4 * km + EdgeProfileLds::kTabDoubles;
  if (!c->edge_dev) RT(rt_malloc((void **)&c->edge_dev, sizeof(double) * n_edge));
  RT(rtf_h2d(c->dp0, dp0, sizeof(double) * km, c->stream));
  // edge_profile coefficients, same arithmetic as nh_utils.F90:1640-1662
  std::vector<double> co(n_edge, 0.);
  double *gk = co.data(), *bet = gk + km, *gam = bet + km, *rbet = gam + km;   // rbet: the fast mode's reciprocals
  const double g0 = dp0[1] / dp0[0];
  c->ec.xt1_top = 2. * g0 * (g0 + 1.);
  c->ec.bet_top = g0 * (g0 + 0.5);
  gam[0] = (1. + g0 * (g0 + 1.5)) / c->ec.bet_top;
  double gkk = 0.;
  for (int k = 2; k <= km; k++) {
    gkk = dp0[k - 2] / dp0[k - 1];
    gk[k - 1] = gkk;
    bet[k - 1] = 2. + 2. * gkk - gam[k - 2];
    gam[k - 1] = gkk / bet[k - 1];
    rbet[k - 1] = 1. / bet[k - 1];
  }
  c->ec.a_bot = 1. + gkk * (gkk + 1.5);
  c->ec.xt1_bot = 2. * gkk * (gkk + 1.);
  c->ec.gk_bot = gkk;
  if (km <= 127) edge_rows(co.data() + 4 * km, km, gk, bet, gam, rbet, c->ec.bet_top, c->ec.a_bot, c->ec.gk_bot);
  RT(rtf_h2d(c->edge_dev, co.data(), sizeof(double) * n_edge, c->stream));
  RT(rtf_sync(c->stream));
  c->ec.gk = c->edge_dev;
  c->ec.bet = c->edge_dev + km;
  c->ec.gam = c->edge_dev + 2 * km;
  c->dp0_ready = true;
  return 0;
}

extern "C" int fv3_update_dz_c(fv3_ctx *c, double dt, const double *zs, const double *ut, const double *vt,
                               const double *gz_in, double *gz, double *ws) {
  if (!c || !c->grid_ready) return fail("fv3_update_dz_c: context has no grid");
  if (!c->dp0_ready) return fail("fv3_update_dz_c: call fv3_set_dp_ref first");
  if (gz_in == gz) return fail("fv3_update_dz_c: gz_in and gz must not alias");
  UpdateDzC kf{c->g, c->g.npz, dt, c->dp0, zs, ut, vt, gz_in, gz, ws};
  RT(launch_p(c, "update_dz_c", col_grid((c->g.nx + 2) * (c->g.ny + 2)), UpdateDzC::lds_doubles(c->g.npz), kf));
  return 0;
}

static NhConsts to_consts(const fv3_ctx *c, const fv3_nh_consts *cn) {
  return NhConsts{cn->grav, cn->rdgas, cn->cp_air, cn->akap, cn->ptop, cn->p_fac, cn->a_imp, c->rff_on ? c->rff_d : nullptr};
}

// the three level tables behind c->rff_d: rff of fast_tau_w_sec, rf and dp of Ray_fast (npz doubles each)
static int level_tables(fv3_ctx *c) {
  if (c->rff_d) return 0;
  const int n = c->g.npz;
  RT(rt_malloc((void **)&c->rff_d, sizeof(double) * 3 * n));
  std::vector<double> one(3 * (size_t)n, 1.0);
  RT(rtf_h2d(c->rff_d, one.data(), sizeof(double) * 3 * n, c->stream));
  RT(rtf_sync(c->stream));
  return 0;
}

extern "C" int fv3_set_fast_tau_w(fv3_ctx *c, int k_rf, const double *rff) {
  if (!c || !c->grid_ready) return fail("fv3_set_fast_tau_w: context has no grid");
  if (k_rf < 0 || k_rf > c->g.npz || (k_rf > 0 && !rff)) return fail("fv3_set_fast_tau_w: k_rf out of range / null table");
  c->rff_on = 0;
  if (k_rf == 0) return 0;
  RT(level_tables(c));
  std::vector<double> t((size_t)c->g.npz, 1.0);
  for (int k = 0; k < k_rf; k++) t[k] = rff[k];
  RT(rtf_sync(c->stream));   // a solver still reading the previous table
  RT(rtf_h2d(c->rff_d, t.data(), sizeof(double) * c->g.npz, c->stream));
  RT(rtf_sync(c->stream));
  c->rff_on = 1;
  return 0;
}

extern "C" int fv3_set_ray_fast(fv3_ctx *c, int kmax, int k_rf, double dm, const double *rf, const double *dp) {
  if (!c || !c->grid_ready) return fail("fv3_set_ray_fast: context has no grid");
  const int n = c->g.npz;
  if (kmax < 0 || kmax > n || k_rf < 0 || k_rf > n || !rf || !dp) return fail("fv3_set_ray_fast: kmax / k_rf out of range or null table");
  if (k_rf > 0 && !(dm > 0.)) return fail("fv3_set_ray_fast: dm (the mass of the levels k <= k_rf) must be positive");
  RT(level_tables(c));
  RT(rtf_sync(c->stream));
  RT(rtf_h2d(c->rff_d + n, rf, sizeof(double) * kmax, c->stream));
  RT(rtf_h2d(c->rff_d + 2 * n, dp, sizeof(double) * n, c->stream));
  RT(rtf_sync(c->stream));
  c->rayf_kmax = kmax; c->rayf_krf = k_rf; c->rayf_dm = dm;
  return 0;
}

extern "C" int fv3_ray_fast(fv3_ctx *c, double *u, double *v, double *w, int hydrostatic) {
  if (!c || !c->grid_ready) return fail("fv3_ray_fast: context has no grid");
  if (c->rayf_kmax < 0) return fail("fv3_ray_fast: call fv3_set_ray_fast first");
  if (!u || !v || (!hydrostatic && !w)) return fail("fv3_ray_fast: null argument");
  const Grid &g = c->g;
  RayFast kf{g, c->rayf_kmax, c->rayf_krf, hydrostatic, c->rayf_dm, c->rff_d + g.npz, c->rff_d + 2 * g.npz, u, v, w};
  RT(launch_c(c, "ray_fast", col_grid((g.nx + 1) * (g.ny + 1)), kf));
  return 0;
}

extern "C" int fv3_mix_dp(fv3_ctx *c, int hydrostatic, double *w, double *delp, double *pt) {
  if (!c || !c->grid_ready) return fail("fv3_mix_dp: context has no grid");
  if (!c->akbk_ready) return fail("fv3_mix_dp: call fv3_set_ak_bk first (dpmin is 1 %% of the reference thickness of a layer)");
  if (!delp || !pt || (!hydrostatic && !w)) return fail("fv3_mix_dp: null argument");
  if (c->g.npz < 2) return fail("fv3_mix_dp: needs npz >= 2");
  const Grid &g = c->g;
  MixDp kf{g, hydrostatic, c->akbk, c->akbk + (g.npz + 1), w, delp, pt};
  RT(launch_c(c, "mix_dp", col_grid(g.nx * g.ny), kf));
  return 0;
}

extern "C" int fv3_set_condensate(fv3_ctx *c, const double *q_con, const double *cappa) {
  if (!c) return fail("fv3_set_condensate: null context");
  c->q_con = q_con;
  c->cappa = cappa;
  return 0;
}

extern "C" int fv3_riem_solver_c(fv3_ctx *c, double dt, const fv3_nh_consts *cn, const double *hs, const double *w3,
                                 const double *pt, const double *delp, double *gz, double *pef, const double *ws) {
  if (!c || !c->grid_ready || !cn) return fail("fv3_riem_solver_c: bad context/arguments");
  if (cn->a_imp <= 0.5) {      // nh_utils.F90:449-459: a_imp < -0.01 SIM3p0_solver, otherwise RIM_2D(c_core = .true.)
    if (c->q_con) return fail("fv3_riem_solver_c: use_cond with a_imp <= 0.5 (SIM3p0 / RIM_2D) is not built");
    if (c->g.npz > kAltKm - 1 || c->g.npz < 2) return fail("fv3_riem_solver_c: SIM3p0 / RIM_2D are built for 2 <= npz <= %d", kAltKm - 1);
    RiemSolverAlt<true> kf{c->g, c->g.npz, cn->a_imp < -0.01 ? 0 : 2, cn->m_split >= 1 ? cn->m_split : 1, dt, to_consts(c, cn), hs, pt, delp, ws,
                           const_cast<double *>(w3), gz, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, pef, 0, 0, 0};
    RT(launch_c(c, "riem_solver_c", col_grid((c->g.nx + 2) * (c->g.ny + 2)), kf));
    return 0;
  }
  if (c->q_con && c->riem_lds && c->g.npz <= 127 && c->g.npz >= 2) {   // use_cond (+ moist_kappa) in the reference's order
    RiemFast<true, true, false, true> kf{c->g, c->g.npz, dt, to_consts(c, cn), hs, pt, delp, ws, const_cast<double *>(w3), gz,
                                         nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, pef, 0, 0, 0, c->q_con, c->cappa};
    RT(launch_p2(c, "riem_solver_c", Dim3{(unsigned)kf.nblocks_x(), (unsigned)kf.nrows(), 1}, kf.kLdsDoubles, kf));
    return 0;
  }
  if (!c->q_con && c->g.npz <= 127 && c->g.npz >= 2) {   // Riem_Solver_c is SIM1 whatever a_imp is
    if (c->riem_lds) {         // the recurrences in the reference's order: the slab kernel's bits
      RiemFast<true, true> kf{c->g, c->g.npz, dt, to_consts(c, cn), hs, pt, delp, ws, const_cast<double *>(w3), gz,
                              nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, pef, 0, 0, 0};
      if (const char *pe_ = std::getenv("FV3_MI355X_RIEM_PROBE")) kf.probe = std::atoi(pe_);
      RT(launch_p2(c, "riem_solver_c", Dim3{(unsigned)kf.nblocks_x(), (unsigned)kf.nrows(), 1}, kf.kLdsDoubles, kf));
      return 0;
    }
  }
  if (need_scratch(c, 4)) return 1;
  const int ncc = (c->g.nx + 2) * (c->g.ny + 2), pool = col_pool(c, ncc);
  if (c->q_con) {
    RiemSolverC<true> kf{c->g, c->g.npz, dt, to_consts(c, cn), hs, w3, pt, delp, ws, gz, pef,
                         c->scratch[0], c->scratch[1], c->scratch[2], c->scratch[3], c->q_con, c->cappa, c->riem_blocked, pool};
    RT(launch_c(c, "riem_solver_c", pool ? col_grid(pool * 256) : col_grid(ncc), kf));
  } else {
    RiemSolverC<false> kf{c->g, c->g.npz, dt, to_consts(c, cn), hs, w3, pt, delp, ws, gz, pef,
                          c->scratch[0], c->scratch[1], c->scratch[2], c->scratch[3], nullptr, nullptr, c->riem_blocked, pool};
    RT(launch_c(c, "riem_solver_c", pool ? col_grid(pool * 256) : col_grid(ncc), kf));
  }
  return 0;
}

extern "C" int fv3_riem_solver3(fv3_ctx *c, double dt, const fv3_nh_consts *cn, const double *zs, double *w,
                                double *delz, const double *pt, const double *delp, double *zh, double *pe, double *ppe,
                                double *pk3, double *pk, double *peln, const double *ws, int use_logp, int last_call,
                                int fp_out) {
  if (!c || !c->grid_ready || !cn) return fail("fv3_riem_solver3: bad context/arguments");
  if (last_call && (!pe || !pk || !peln)) return fail("fv3_riem_solver3: last_call needs pe, pk, peln");
  if (cn->a_imp <= 0.5) {      // nh_core.F90:169-177: a_imp < -0.999 SIM3p0_solver, < -0.5 SIM3_solver, otherwise RIM_2D
    if (c->q_con || c->cappa) return fail("fv3_riem_solver3: use_cond / moist_kappa with a_imp <= 0.5 (SIM3 / SIM3p0 / RIM_2D) is not built");
    if (c->g.npz > kAltKm - 1 || c->g.npz < 2) return fail("fv3_riem_solver3: SIM3 / SIM3p0 / RIM_2D are built for 2 <= npz <= %d", kAltKm - 1);
    RiemSolverAlt<false> kf{c->g, c->g.npz, cn->a_imp < -0.999 ? 0 : (cn->a_imp < -0.5 ? 1 : 2), cn->m_split >= 1 ? cn->m_split : 1, dt,
                            to_consts(c, cn), zs, pt, delp, ws, w, zh, delz, ppe, pk3, pe, pk, peln, nullptr, use_logp, last_call, fp_out};
    RT(launch_c(c, "riem_solver3", col_grid(c->g.nx * c->g.ny), kf));
    return 0;
  }
  if ((c->q_con || c->cappa) && c->riem_lds && c->g.npz <= 127 && c->g.npz >= 2) {   // use_cond / moist_kappa, SIM1 or SIM
    if (cn->a_imp > 0.999) {
      RiemFast<false, true, false, true> kf{c->g, c->g.npz, dt, to_consts(c, cn), zs, pt, delp, ws, w, zh, delz, ppe, pk3, pe, pk, peln, nullptr,
                                            use_logp, last_call, fp_out, c->q_con, c->cappa};
      RT(launch_p2(c, "riem_solver3", Dim3{(unsigned)kf.nblocks_x(), (unsigned)kf.nrows(), 1}, kf.kLdsDoubles, kf));
    } else {
      RiemFast<false, true, true, true> kf{c->g, c->g.npz, dt, to_consts(c, cn), zs, pt, delp, ws, w, zh, delz, ppe, pk3, pe, pk, peln, nullptr,
                                           use_logp, last_call, fp_out, c->q_con, c->cappa};
      RT(launch_p2(c, "riem_solver3", Dim3{(unsigned)kf.nblocks_x(), (unsigned)kf.nrows(), 1}, kf.kLdsDoubles, kf));
    }
    return 0;
  }
  if (!c->q_con && !c->cappa && cn->a_imp <= 0.999 && c->riem_lds && c->g.npz <= 127 && c->g.npz >= 2) {   // SIM_solver (the reference's default a_imp = 0.75)
    RiemFast<false, true, true> kf{c->g, c->g.npz, dt, to_consts(c, cn), zs, pt, delp, ws, w, zh, delz, ppe, pk3, pe, pk, peln, nullptr,
                                   use_logp, last_call, fp_out};
    RT(launch_p2(c, "riem_solver3", Dim3{(unsigned)kf.nblocks_x(), (unsigned)kf.nrows(), 1}, kf.kLdsDoubles, kf));
    return 0;
  }
  if (!c->q_con && !c->cappa && cn->a_imp > 0.999 && c->g.npz <= 127 && c->g.npz >= 2) {
    if (c->riem_lds) {
      RiemFast<false, true> kf{c->g, c->g.npz, dt, to_consts(c, cn), zs, pt, delp, ws, w, zh, delz, ppe, pk3, pe, pk, peln, nullptr,
                               use_logp, last_call, fp_out};
      if (const char *pe_ = std::getenv("FV3_MI355X_RIEM_PROBE")) kf.probe = std::atoi(pe_);
      RT(launch_p2(c, "riem_solver3", Dim3{(unsigned)kf.nblocks_x(), (unsigned)kf.nrows(), 1}, kf.kLdsDoubles, kf));
      return 0;
    }
  }
  if (need_scratch(c, 4)) return 1;
  const int ncc = c->g.nx * c->g.ny, pool = col_pool(c, ncc);
  if (c->q_con || c->cappa) {
    RiemSolver3<true> kf{c->g, c->g.npz, dt, to_consts(c, cn), zs, pt, delp, ws, w, delz, zh, pe, ppe, pk3, pk, peln,
                         use_logp, last_call, fp_out, c->scratch[0], c->scratch[1], c->scratch[2], c->scratch[3],
                         c->q_con, c->cappa, c->riem_blocked, pool};
    RT(launch_c(c, "riem_solver3", pool ? col_grid(pool * 256) : col_grid(ncc), kf));
  } else {
    RiemSolver3<false> kf{c->g, c->g.npz, dt, to_consts(c, cn), zs, pt, delp, ws, w, delz, zh, pe, ppe, pk3, pk, peln,
                          use_logp, last_call, fp_out, c->scratch[0], c->scratch[1], c->scratch[2], c->scratch[3],
                          nullptr, nullptr, c->riem_blocked, pool};
    RT(launch_c(c, "riem_solver3", pool ? col_grid(pool * 256) : col_grid(ncc), kf));
  }
  return 0;
}

extern "C" int fv3_update_dz_d(fv3_ctx *c, int hord, const double *zs, const double *zh_in, double *zh_out,
                               const double *crx, const double *cry, const double *xfx, const double *yfx, double *ws,
                               double rdt) {
  if (!c || !c->grid_ready) return fail("fv3_update_dz_d: context has no grid");
  if (!c->dp0_ready) return fail("fv3_update_dz_d: call fv3_set_dp_ref first");
  if (!c->lev_ready) return fail("fv3_update_dz_d: call fv3_dsw_levels_upload first");
  if (!tp_ord_supported(hord)) return fail("fv3_update_dz_d: hord=%d not supported", hord);
  if (zh_in == zh_out) return fail("fv3_update_dz_d: zh_in and zh_out must not alias");
  if (need_scratch(c, 4)) return 1;  // crx_adv, xfx_adv (CX x (km+1)), cry_adv, yfx_adv (CY x (km+1)) fit in A x (km+1)
  const Grid &g = c->g;
  const int km = g.npz;
  double *cxa = c->scratch[0], *xfa = c->scratch[1], *cya = c->scratch[2], *yfa = c->scratch[3];
  if (c->riem_lds && km >= 3 && km <= 127) {   // levels across the lanes, the elimination in the reference's order: the slab kernel's bits
    EdgeProfileLds kf{g, km, c->ec, c->edge_dev + 3 * km, crx, xfx, cxa, xfa, (int)g.nCX(), cry, yfx, cya, yfa, (int)g.nCY(),
                      c->edge_dev + 4 * km};
    RT(launch_p2(c, "edge_profile", Dim3{(unsigned)kf.nblocks(), 1, 1}, 2 * kFBuf, kf));
  } else {
    EdgeProfile kf{g, km, c->ec, crx, xfx, cxa, xfa, (int)g.nCX(), cry, yfx, cya, yfa, (int)g.nCY()};
    RT(launch_c(c, "edge_profile", col_grid((int)(g.nCX() + g.nCY())), kf));
  }
  if (is_cubed(c)) {
    double *fx = cs_scratch(c, 8), *fy = cs_scratch(c, 9);
    if (!fx || !fy) return fail("fv3_update_dz_d: out of device memory");
    // Hybrid (see dsw_cubed): the marching transport of the interface heights over the whole face, then the cubed fv_tp_2d
    // passes on the frame along the face edges (zh_out is not an input: the passes simply overwrite the frame)
    const int wo = c->cubed_frame, wm = wo + c->cubed_reach;
    const bool hyb = c->use_march && wo > 0 && g.npx == g.npy && g.npx - 1 >= 2 * wm + 8;
    if (hyb && c->n_plain_z > 0) {
      MarchDims md = make_march_dims(g, seg_rows(c, c->march_tj, g.npz));
      md.klist = c->klist_z;
      const int nwz = md.nwaves(c->n_plain_z);
      RT(dispatch_hord(hord, [&](auto H) {
        ZhMarch<decltype(H)::value> kf{g, md, zh_in, cxa, cya, xfa, yfa, zh_out};
        return launch_w(c, "zh_transport", nwz, kf);
      }));
    }
    // the plain levels (no damping: klist_z[0 : n_plain_z)): frames when the marching kernel took the interior; the levels
    // with del6_vt_flux damping (nh_utils.F90:268-284): passes on the whole face + the damping fluxes (cubed_damp.h)
    if (c->n_plain_z > 0) {
      const PassRegion rm{hyb ? wm : 0, c->klist_z, c->n_plain_z}, ro{hyb ? wo : 0, c->klist_z, c->n_plain_z};
      if (frame_fused_on()) {
        const TpfField fl[3] = {TpfField{zh_in, fx, fy, hord}, TpfField{}, TpfField{}};
        RT(tp2d_frame_fused(c, fl, 1, cxa, cya, xfa, yfa, wo + 1, c->klist_z, c->n_plain_z, "zhc_tp", !hyb));
      } else if (tp2d_cubed(c, km + 1, zh_in, cxa, cya, hord, fx, fy, xfa, yfa, nullptr, nullptr, nullptr, nullptr, "zhc_tp", &rm)) {
        return 1;
      }
      RT(launch_pass(c, "zhc_fin", g.is, g.ie, g.js, g.je, ro, ZhCubedFinal{g, zh_in, fx, fy, xfa, yfa, zh_out}));
    }
    if (c->n_damp_z > 0) {
      const PassRegion rd{0, c->klist_z + c->n_plain_z, c->n_damp_z};
      if (frame_fused_on()) {
        const TpfField fl[3] = {TpfField{zh_in, fx, fy, hord}, TpfField{}, TpfField{}};
        RT(tp2d_frame_fused(c, fl, 1, cxa, cya, xfa, yfa, 5, rd.klist, rd.nk, "zhc_tp", true));
      } else if (tp2d_cubed(c, km + 1, zh_in, cxa, cya, hord, fx, fy, xfa, yfa, nullptr, nullptr, nullptr, nullptr, "zhc_tp", &rd)) {
        return 1;
      }
      DelnCubedState d;
      d.g = g; d.q = zh_in; d.mass = nullptr; d.fx = d.fy = nullptr; d.nord = c->lev_ext_i; d.coef = c->lev_ext_d; d.thresh = 1.E-5;
      d.corner_area = 2;
      d.d2 = cs_scratch(c, 4); d.fx2 = cs_scratch(c, 10); d.fy2 = cs_scratch(c, 11);
      if (!d.d2 || !d.fx2 || !d.fy2) return fail("fv3_update_dz_d: out of device memory");
      RT(launch_pass(c, "zhc_del6", g.isd, g.ied, g.jsd, g.jed, rd, DelnCubedL1{d}));
      RT(launch_pass(c, "zhc_del6", g.isd, g.ied + 1, g.jsd, g.jed + 1, rd, DelnCubedL24{d, 1, 0}));
      for (int n = 1; n <= c->lev_max_nord_v; n++) {
        RT(launch_pass(c, "zhc_del6", g.isd, g.ied, g.jsd, g.jed, rd, DelnCubedL3{d, n}));
        RT(launch_pass(c, "zhc_del6", g.isd, g.ied + 1, g.jsd, g.jed + 1, rd, DelnCubedL24{d, 0, n}));
      }
      ZhCubedFinal kf{g, zh_in, fx, fy, xfa, yfa, zh_out};
      kf.damp = c->lev_ext_d; kf.fx2 = d.fx2; kf.fy2 = d.fy2;
      RT(launch_pass(c, "zhc_fin", g.is, g.ie, g.js, g.je, rd, kf));
    }
    ZhLimit kf{g, km, rdt, zs, zh_out, ws};
    RT(launch_c(c, "zh_limit", col_grid(g.nx * g.ny), kf));
    return 0;
  }
  constexpr int TI = FV3_DSW_TI, TJ = FV3_DSW_TJ;
  const bool march = c->use_march != 0;
  if (march && c->n_plain_z > 0) {
    MarchDims md = make_march_dims(g, seg_rows(c, c->march_tj, g.npz));
    md.klist = c->klist_z;
    const int nwz = md.nwaves(c->n_plain_z);
    RT(dispatch_hord(hord, [&](auto H) {
      ZhMarch<decltype(H)::value> kf{g, md, zh_in, cxa, cya, xfa, yfa, zh_out};
      return launch_w(c, "zh_transport", nwz, kf);
    }));
  }
  if (!march || c->n_damp_z > 0) {
    ZhTransport<TI, TJ> kf{g, hord, zh_in, cxa, cya, xfa, yfa, c->lev_ext_i, c->lev_ext_d, zh_out,
                           march ? c->klist_z + c->n_plain_z : nullptr};
    Dim3 grid;
    grid.x = (unsigned)((g.nx + TI - 1) / TI);
    grid.y = (unsigned)((g.ny + TJ - 1) / TJ);
    grid.z = (unsigned)(march ? c->n_damp_z : km + 1);
    RT(launch_p(c, "zh_transport", grid, ZhTransport<TI, TJ>::lds_doubles, kf));
  }
  {
    ZhLimit kf{g, km, rdt, zs, zh_out, ws};
    RT(launch_c(c, "zh_limit", col_grid(g.nx * g.ny), kf));
  }
  return 0;
}

extern "C" int fv3_p_grad_c(fv3_ctx *c, double dt2, const double *delpc, const double *pkc, const double *gz,
                            double *uc, double *vc, int hydrostatic) {
  if (!c || !c->grid_ready) return fail("fv3_p_grad_c: context has no grid");
  const Grid &g = c->g;
  PGradC kf{g, dt2, hydrostatic, delpc, pkc, gz, uc, vc};
  RT(launch_c(c, "p_grad_c", col_grid(kf.ncol()), kf));
  return 0;
}

extern "C" int fv3_pt_to_theta_v(fv3_ctx *c, int hydrostatic, double zvir, double kappa, double rdgas, double grav,
                                 double *pt, const double *delp, const double *delz, const double *qv, double *pkz) {
  if (!c || !c->grid_ready) return fail("fv3_pt_to_theta_v: context has no grid");
  if (!pt || !pkz || (hydrostatic <= 0 && (!delp || !delz))) return fail("fv3_pt_to_theta_v: null field");
  if (hydrostatic < -1 || hydrostatic > 1) return fail("fv3_pt_to_theta_v: hydrostatic must be 0, 1 or -1 (pkz only)");
  const Grid &g = c->g;
  RemapPar mp{};
  const double *mq = nullptr;
  if (c->moist_on && (c->moist.moist_kappa || c->moist.use_cond)) {
    const fv3_moist_params &m = c->moist;
    if (m.use_cond && !c->moist_qcon) return fail("fv3_pt_to_theta_v: use_cond needs q_con (fv3_set_moist)");
    mp.use_cond = m.use_cond;
    mp.q_con = c->moist_qcon;
    mp.cappa = c->moist_cappa;
    if (m.moist_kappa && hydrostatic <= 0) {
      if (!qv || m.sphum < 1 || !c->moist_qcon || !c->moist_cappa)
        return fail("fv3_pt_to_theta_v: moist_kappa needs qv = &q(.,.,1,sphum), sphum, q_con and cappa (fv3_set_moist)");
      mp.moist_kappa = 1;
      mp.nwat = m.nwat; mp.sphum = m.sphum; mp.liq_wat = m.liq_wat; mp.rainwat = m.rainwat; mp.ice_wat = m.ice_wat;
      mp.snowwat = m.snowwat; mp.graupel = m.graupel;
      mp.cv_vap = m.cv_vap; mp.c_liq = m.c_liq; mp.c_ice = m.c_ice;
      mp.rdgas = rdgas; mp.cv_air = rdgas / kappa - rdgas;   // cp_air - rdgas with cp_air = rdgas / kappa (constants_mod)
      mq = qv - (size_t)(m.sphum - 1) * g.nA() * g.npz;
    }
  }
  PtToThetaV kf{g, hydrostatic, zvir, kappa, -rdgas / grav, pt, delp, delz, qv, pkz, mp, mq};
  Dim3 grid;
  grid.x = (unsigned)((g.nx * g.ny + PtToThetaV::CH - 1) / PtToThetaV::CH);
  grid.y = 1;
  grid.z = (unsigned)g.npz;
  RT(launch_p(c, "pt_to_theta_v", grid, 0, kf));
  return 0;
}

extern "C" int fv3_c2l(fv3_ctx *c, int ord, const double *u, const double *v, double *ua, double *va) {
  if (!c || !c->grid_ready) return fail("fv3_c2l: context has no grid");
  if (!u || !v || !ua || !va) return fail("fv3_c2l: null field");
  if (ord != 2 && ord != 4) return fail("fv3_c2l: c2l_ord must be 2 or 4");
  const Grid &g = c->g;
  if (g.grid_type == 3) return fail("fv3_c2l: grid_type 3 is not built");
  if (g.grid_type < 3 && !(c->cg.ready && c->cg.a11)) return fail("fv3_c2l: cubed-sphere face without a11 .. a22 (fv3_grid_upload_cubed)");
  C2L kf{g, ord, 1, u, v, nullptr, ua, va, nullptr, c->cg};
  Dim3 grid;
  grid.x = (unsigned)((g.nx * g.ny + C2L::CH - 1) / C2L::CH);
  grid.y = 1;
  grid.z = (unsigned)g.npz;
  RT(launch_p(c, "c2l", grid, 0, kf));
  return 0;
}

extern "C" int fv3_rayleigh_u2f(fv3_ctx *c, int kmax, int hydrostatic, const double *u, const double *v,
                                const double *w, double *ua, double *va, double *u2f) {
  if (!c || !c->grid_ready) return fail("fv3_rayleigh_u2f: context has no grid");
  if (!u || !v || !ua || !va || !u2f || (!hydrostatic && !w)) return fail("fv3_rayleigh_u2f: null field");
  const Grid &g = c->g;
  if (g.grid_type == 3) return fail("fv3_rayleigh_u2f: grid_type 3 is not built");
  if (g.grid_type < 3 && !(c->cg.ready && c->cg.a11)) return fail("fv3_rayleigh_u2f: cubed-sphere face without a11 .. a22");
  if (kmax < 0 || kmax > g.npz) return fail("fv3_rayleigh_u2f: kmax out of range");
  if (kmax == 0) return 0;
  C2L kf{g, 2, hydrostatic, u, v, w, ua, va, u2f, c->cg};
  Dim3 grid;
  grid.x = (unsigned)((g.nx * g.ny + C2L::CH - 1) / C2L::CH);
  grid.y = 1;
  grid.z = (unsigned)kmax;
  RT(launch_p(c, "rayleigh_u2f", grid, 0, kf));
  return 0;
}

extern "C" int fv3_rayleigh_apply(fv3_ctx *c, int kmax, int conserve, int hydrostatic, double cp, double rg, double ptop,
                                  const double *pm, const double *rf, const double *u2f, double *pt, double *delz,
                                  double *u, double *v, double *w) {
  if (!c || !c->grid_ready) return fail("fv3_rayleigh_apply: context has no grid");
  if (!pm || !rf || !u2f || !pt || !u || !v || (!hydrostatic && (!w || !delz)))
    return fail("fv3_rayleigh_apply: null argument");
  const Grid &g = c->g;
  if (kmax < 0 || kmax > g.npz) return fail("fv3_rayleigh_apply: kmax out of range");
  if (kmax == 0) return 0;
  if (!c->ray_d) RT(rt_malloc((void **)&c->ray_d, sizeof(double) * 2 * g.npz));
  RT(rtf_h2d(c->ray_d, pm, sizeof(double) * kmax, c->stream));
  RT(rtf_h2d(c->ray_d + g.npz, rf, sizeof(double) * kmax, c->stream));
  RT(rtf_sync(c->stream));  // pm / rf are the caller's host arrays
  RayleighApply kf{g, conserve, hydrostatic, cp, rg, ptop, c->ray_d, c->ray_d + g.npz, u2f, pt, delz, u, v, w};
  Dim3 grid;
  grid.x = (unsigned)(((g.nx + 1) * (g.ny + 1) + RayleighApply::CH - 1) / RayleighApply::CH);
  grid.y = 1;
  grid.z = (unsigned)kmax;
  RT(launch_p(c, "rayleigh_apply", grid, 0, kf));
  return 0;
}

extern "C" int fv3_rayleigh_super(fv3_ctx *c, int kmax, int conserve, int hydrostatic, double cp, double rg, double ptop,
                                  const double *pm, const double *rf, const double *ua, const double *va, double *pt,
                                  double *u, double *v, double *w, const double *u00, const double *v00) {
  if (!c || !c->grid_ready) return fail("fv3_rayleigh_super: context has no grid");
  if (!pm || !rf || !ua || !va || !pt || !u || !v || (!hydrostatic && !w)) return fail("fv3_rayleigh_super: null argument");
  if ((u00 == nullptr) != (v00 == nullptr)) return fail("fv3_rayleigh_super: u00 and v00 must be given together");
  const Grid &g = c->g;
  if (kmax < 0 || kmax > g.npz) return fail("fv3_rayleigh_super: kmax out of range");
  if (kmax == 0) return 0;
  if (!c->ray_d) RT(rt_malloc((void **)&c->ray_d, sizeof(double) * 2 * g.npz));
  RT(rtf_h2d(c->ray_d, pm, sizeof(double) * kmax, c->stream));
  RT(rtf_h2d(c->ray_d + g.npz, rf, sizeof(double) * kmax, c->stream));
  RT(rtf_sync(c->stream));  // pm / rf are the caller's host arrays
  RayleighSuper kf{g, conserve, hydrostatic, cp, rg, ptop, c->ray_d, c->ray_d + g.npz, ua, va, pt, u, v, w, u00, v00};
  Dim3 grid;
  grid.x = (unsigned)(((g.nx + 1) * (g.ny + 1) + RayleighSuper::CH - 1) / RayleighSuper::CH);
  grid.y = 1;
  grid.z = (unsigned)kmax;
  RT(launch_p(c, "rayleigh_super", grid, 0, kf));
  return 0;
}

extern "C" int fv3_compute_aam(fv3_ctx *c, double radius, double omega, double agrav, double ptop, const double *coslat, const double *ua,
                               const double *delp, double *aam, double *m_fac, double *ps) {
  if (!c || !c->grid_ready) return fail("fv3_compute_aam: context has no grid");
  if (!coslat || !ua || !delp || !aam || !m_fac || !ps) return fail("fv3_compute_aam: null argument");
  AamColumns kf{c->g, radius, omega, agrav, ptop, coslat, ua, delp, aam, m_fac, ps};
  RT(launch_c(c, "compute_aam", col_grid(c->g.nx * c->g.ny), kf));
  return 0;
}

extern "C" int fv3_consv_am_apply(fv3_ctx *c, double u00, const double *l2c_u, const double *l2c_v, double *u, double *v) {
  if (!c || !c->grid_ready) return fail("fv3_consv_am_apply: context has no grid");
  if (!l2c_u || !l2c_v || !u || !v) return fail("fv3_consv_am_apply: null argument");
  const Grid &g = c->g;
  ConsvAmApply kf{g, u00, l2c_u, l2c_v, u, v};
  Dim3 grid;
  grid.x = (unsigned)(((g.nx + 1) * (g.ny + 1) + ConsvAmApply::CH - 1) / ConsvAmApply::CH);
  grid.y = 1;
  grid.z = (unsigned)g.npz;
  RT(launch_p(c, "consv_am_apply", grid, 0, kf));
  return 0;
}

extern "C" int fv3_heat_source_accum(fv3_ctx *c, double *heat_source, const double *heat_s) {
  if (!c || !c->grid_ready || !heat_source || !heat_s) return fail("fv3_heat_source_accum: bad context/arguments");
  const Grid &g = c->g;
  HeatAccum kf{g, heat_source, heat_s};
  Dim3 grid;
  grid.x = (unsigned)((g.nx * g.ny + HeatAccum::CH - 1) / HeatAccum::CH);
  grid.y = 1;
  grid.z = (unsigned)g.npz;
  RT(launch_p(c, "heat_accum", grid, 0, kf));
  return 0;
}

extern "C" int fv3_del2_cubed(fv3_ctx *c, double *q, int nk, double cd, int nmax) {
  if (!c || !c->grid_ready || !q) return fail("fv3_del2_cubed: bad context/arguments");
  if (nk < 1 || nk > c->g.npz + 1) return fail("fv3_del2_cubed: nk out of range");
  if (need_scratch(c, 1)) return 1;
  const Grid &g = c->g;
  const int ntimes = nmax < 3 ? nmax : 3;
  double *src = q, *dst = c->scratch[0];
  for (int n = 1; n <= ntimes; n++) {
    Del2Pass kf{g, src, dst, cd, ntimes - n};
    Dim3 grid;
    grid.x = (unsigned)((g.nid * g.njd + Del2Pass::CH - 1) / Del2Pass::CH);
    grid.y = 1;
    grid.z = (unsigned)nk;
    RT(launch_p(c, "del2_cubed", grid, 0, kf));
    double *t = src; src = dst; dst = t;
  }
  if (src != q) RT(grp_stream_op(c, 2, q, src, sizeof(double) * g.nA() * nk, 0));
  return 0;
}

extern "C" int fv3_apply_heat_source(fv3_ctx *c, int n_con, int hydrostatic, double bdt, double delt_max, double cp_air,
                                     double cv_air, double rdgas, double grav, double *pt, double *heat_source,
                                     const double *delp, const double *delz, double *pkz) {
  if (!c || !c->grid_ready) return fail("fv3_apply_heat_source: context has no grid");
  if (!pt || !heat_source || !delp || !pkz || (!hydrostatic && !delz)) return fail("fv3_apply_heat_source: null field");
  const Grid &g = c->g;
  if (n_con < 0 || n_con > g.npz) return fail("fv3_apply_heat_source: n_con out of range");
  if (n_con == 0) return 0;
  // moist_kappa: the cappa of fv3_set_condensate (the array the Riemann solvers use) gives the exponent of pkz
  HeatApply kf{g, n_con, hydrostatic, bdt, delt_max, cp_air, cv_air, -rdgas / grav, rdgas / cv_air, pt, heat_source, delp,
               delz, pkz, hydrostatic ? nullptr : c->cappa};
  Dim3 grid;
  grid.x = (unsigned)((g.nx * g.ny + HeatApply::CH - 1) / HeatApply::CH);
  grid.y = 1;
  grid.z = (unsigned)n_con;
  RT(launch_p(c, "heat_apply", grid, 0, kf));
  return 0;
}

extern "C" int fv3_zh_from_delz(fv3_ctx *c, const double *zs, const double *delz, double *zh) {
  if (!c || !c->grid_ready) return fail("fv3_zh_from_delz: context has no grid");
  ZhFromDelz kf{c->g, c->g.npz, zs, delz, zh};
  RT(launch_c(c, "zh_from_delz", col_grid(c->g.nx * c->g.ny), kf));
  return 0;
}

// a2b_ord4 of up to four fields: the LDS-tile kernel (grid_type >= 3) or the cubed-sphere passes (scratch 0..7)
template <int TI, int TJ>
static int run_a2b(fv3_ctx *c, const A2BCorners<TI, TJ> &kf, int nlev_max, const char *who) {
  const Grid &g = c->g;
  // the label follows the caller (reference_timer maps labels to the reference's timers by prefix)
  const std::string l0 = who ? std::string(who) : std::string("a2b_corners"), la = who ? l0 + "_pa" : std::string("a2bc_pa"),
                    lb = who ? l0 + "_pb" : std::string("a2bc_pb");
  static std::vector<std::string> keep;   // labels outlive the launch (profiling records point at them)
  auto lab = [&](const std::string &x) -> const char * {
    for (const auto &k : keep) if (k == x) return k.c_str();
    keep.reserve(64);
    keep.push_back(x);
    return keep.back().c_str();
  };
  if (is_cubed(c)) {
    if (!c->cg.ready) return fail("a2b_ord4: cubed-sphere context without fv3_grid_upload_cubed");
    A2bCubedState s;
    s.g = g; s.cg = c->cg; s.nf = kf.nf; s.override_mask = kf.override_mask;
    for (int f = 0; f < 4; f++) {
      s.in[f] = kf.in[f]; s.out[f] = kf.out[f]; s.nlev[f] = kf.nlev[f]; s.scale[f] = kf.scale[f]; s.top[f] = kf.top[f];
      s.qx[f] = s.qy[f] = nullptr;
      if (f < kf.nf) {
        if (!(s.qx[f] = cs_scratch(c, 2 * f)) || !(s.qy[f] = cs_scratch(c, 2 * f + 1))) return fail("a2b_ord4: out of device memory");
      }
    }
    // Hybrid: away from the face edges a2b_ord4 is the 4th-order form of the LDS-tile kernel (the one-sided forms touch
    // qx at i <= 2, qy at j <= 2 and the corners next to them), so that kernel takes the whole face first and the passes
    // then rewrite a frame of 4 points (qx, qy two points wider).  Outputs and inputs are distinct arrays: no masks needed.
    const int wa = c->cubed_frame ? 4 : 0;
    const bool hyb = wa > 0 && g.npx == g.npy && g.npx - 1 >= 2 * (wa + 3) + 8;
    if (hyb) {
      Dim3 grid;
      grid.x = (unsigned)((g.nx + 1 + TI - 1) / TI);
      grid.y = (unsigned)((g.ny + 1 + TJ - 1) / TJ);
      grid.z = (unsigned)nlev_max;
      A2BCorners<TI, TJ> kc = kf;
      kc.sum_form = 1;
      RT(launch_p(c, lab(l0), grid, A2BCorners<TI, TJ>::lds_doubles, kc));
    }
    RT(launch_pass(c, lab(la), 1, g.npx, 1, g.npy, PassRegion{hyb ? wa + 3 : 0, nullptr, nlev_max}, A2bCubedPa{s}));
    RT(launch_pass(c, lab(lb), 1, g.npx, 1, g.npy, PassRegion{hyb ? wa : 0, nullptr, nlev_max}, A2bCubedPb{s}));
    return 0;
  }
  Dim3 grid;
  grid.x = (unsigned)((g.nx + 1 + TI - 1) / TI);
  grid.y = (unsigned)((g.ny + 1 + TJ - 1) / TJ);
  grid.z = (unsigned)nlev_max;
  return launch_p(c, lab(l0), grid, A2BCorners<TI, TJ>::lds_doubles, kf);
}

static int nh_p_grad_impl(fv3_ctx *c, double *u, double *v, const double *pp, const double *gz, double gz_scale, const double *delp,
                          const double *pk, double dt, double top_value, double beta, double *du, double *dv);
extern "C" int fv3_nh_p_grad(fv3_ctx *c, double *u, double *v, const double *pp, const double *gz, double gz_scale,
                             const double *delp, const double *pk, double dt, double top_value) {
  return nh_p_grad_impl(c, u, v, pp, gz, gz_scale, delp, pk, dt, top_value, 0., nullptr, nullptr);
}
extern "C" int fv3_split_p_grad(fv3_ctx *c, double *u, double *v, const double *pp, const double *gz, double gz_scale,
                                const double *delp, const double *pk, double beta, double dt, double top_value, double *du, double *dv) {
  if (!du || !dv) return fail("fv3_split_p_grad: du, dv (U / V x npz, zero before the first call) are required");
  return nh_p_grad_impl(c, u, v, pp, gz, gz_scale, delp, pk, dt, top_value, beta, du, dv);
}
static int nh_p_grad_impl(fv3_ctx *c, double *u, double *v, const double *pp, const double *gz, double gz_scale, const double *delp,
                          const double *pk, double dt, double top_value, double beta, double *du, double *dv) {
  if (!c || !c->grid_ready) return fail("fv3_nh_p_grad: context has no grid");
  const Grid &g = c->g;
  const int km = g.npz;
  constexpr int TI = 32, TJ = 16;
  if (c->pgrad_fused && !du && !is_cubed(c)) {     // the corner values stay in LDS
    // 32 x 8 corners per workgroup: one point per thread, 114 registers, 25 KB of LDS -- four workgroups per CU.  Measured at C384 L127
    // (tools/a2b_ab.py, same arrays): 32 x 16 tiles (two points per thread, 164 registers, 43 KB: three per CU) 0.480 ms, 32 x 8 0.403;
    // a2b_ord4 + the gradient as two kernels 0.29 + 0.32
    constexpr int FI = 32, FJ = 8;
    NhPGradFused<FI, FJ> kf{g, dt, gz_scale, top_value, pp, pk, gz, delp, u, v};
    Dim3 grid;
    grid.x = (unsigned)((g.nx + 1 + FI - 1) / FI);
    grid.y = (unsigned)((g.ny + 1 + FJ - 1) / FJ);
    grid.z = (unsigned)kf.nchunks();
    return launch_p(c, "nh_p_grad", grid, NhPGradFused<FI, FJ>::lds_doubles, kf);
  }
  if (need_scratch(c, 4)) return 1;
  {
    A2BCorners<TI, TJ> kf;
    kf.g = g;
    kf.in[0] = pp; kf.in[1] = pk; kf.in[2] = gz; kf.in[3] = delp;
    for (int f = 0; f < 4; f++) kf.out[f] = c->scratch[f];
    kf.nlev[0] = kf.nlev[1] = kf.nlev[2] = km + 1;
    kf.nlev[3] = km;
    kf.nf = 4;
    kf.scale[0] = kf.scale[1] = kf.scale[3] = 1.0;
    kf.scale[2] = gz_scale;
    kf.top[0] = 0.; kf.top[1] = top_value; kf.top[2] = kf.top[3] = 0.;
    kf.override_mask = 3;
    RT((run_a2b<TI, TJ>(c, kf, km + 1)));
  }
  {
    if (du) {
      NhPGrad<true> kf{g, dt, c->scratch[0], c->scratch[1], c->scratch[2], c->scratch[3], u, v, beta, du, dv};
      RT(launch_c(c, "nh_p_grad", col_grid(kf.ncol()), kf));
    } else {
      NhPGrad<false> kf{g, dt, c->scratch[0], c->scratch[1], c->scratch[2], c->scratch[3], u, v};
      RT(launch_c(c, "nh_p_grad", col_grid(kf.ncol()), kf));
    }
  }
  return 0;
}

extern "C" int fv3_omga_update(fv3_ctx *c, double rdt, double ptop, const double *pe, const double *delp_before,
                               double *omga) {
  if (!c || !c->grid_ready || !pe || !delp_before || !omga) return fail("fv3_omga_update: bad context/arguments");
  OmgaUpdate kf{c->g, c->g.npz, rdt, ptop, pe, delp_before, omga};
  RT(launch_c(c, "omga_update", col_grid(c->g.nx * c->g.ny), kf));
  return 0;
}

extern "C" int fv3_adv_pe(fv3_ctx *c, double ptop, const double *ua, const double *va, const double *delp_before, double *omga) {
  if (!c || !c->grid_ready || !ua || !va || !delp_before || !omga) return fail("fv3_adv_pe: bad context/arguments");
  const Grid &g = c->g;
  if (g.grid_type >= 3) return fail("fv3_adv_pe: en1 / en2 are not defined for grid_type >= 3 (fv_grid_utils.F90:628)");
  if (!(c->cg.ready && c->cg.ec1)) return fail("fv3_adv_pe: cubed-sphere face without ec1 .. en2 (fv3_grid_upload_cubed)");
  if (need_scratch(c, 2)) return 1;
  PemColumns kp{g, g.npz, ptop, delp_before, c->scratch[0]};
  RT(launch_c(c, "adv_pe_pem", col_grid((g.nx + 2) * (g.ny + 2)), kp));
  Dim3 grid;
  grid.y = 1;
  grid.z = (unsigned)g.npz;
  AdvPeCorners kc{g, c->cg, c->scratch[0], c->scratch[1]};
  grid.x = (unsigned)(((g.nx + 1) * (g.ny + 1) + AdvPeCorners::CH - 1) / AdvPeCorners::CH);
  RT(launch_p(c, "adv_pe_corners", grid, 0, kc));
  AdvPe kf{g, c->cg, g.npz, ua, va, c->scratch[1], omga};
  grid.x = (unsigned)((g.nx * g.ny + AdvPe::CH - 1) / AdvPe::CH);
  RT(launch_p(c, "adv_pe", grid, 0, kf));
  return 0;
}

extern "C" int fv3_divg2_ext(fv3_ctx *c, double d_ext, const double *delp, const double *vt, double *divg2) {
  if (!c || !c->grid_ready || !delp || !vt || !divg2) return fail("fv3_divg2_ext: bad context/arguments");
  const Grid &g = c->g;
  RT(grp_stream_op(c, 1, divg2, nullptr, sizeof(double) * g.nA(), 0));
  if (!(d_ext > 0.)) return 0;
  if (is_cubed(c) && !c->cg.ready) return fail("fv3_divg2_ext: cubed-sphere context without fv3_grid_upload_cubed");
  Divg2Ext kf{g, g.npz, d_ext * g.da_min_c, delp, vt, divg2, c->cg};
  RT(launch_c(c, "divg2_ext", col_grid((g.nx + 1) * (g.ny + 1)), kf));
  return 0;
}

static int one_grad_p_impl(fv3_ctx *c, double *u, double *v, const double *pk, const double *gz, const double *divg2, double dt,
                           double ptk, double beta, double *du, double *dv, const double *delp = nullptr, double gz_scale = 1.0);
extern "C" int fv3_one_grad_p(fv3_ctx *c, double *u, double *v, const double *pk, const double *gz, const double *divg2,
                              double dt, double ptk) {
  return one_grad_p_impl(c, u, v, pk, gz, divg2, dt, ptk, 0., nullptr, nullptr);
}
extern "C" int fv3_one_grad_p_nh(fv3_ctx *c, double *u, double *v, const double *pk, const double *gz, const double *divg2,
                                 const double *delp, double dt, double ptop, double gz_scale) {
  if (!delp) return fail("fv3_one_grad_p_nh: delp is required (the layer weights are a2b_ord4 of delp)");
  return one_grad_p_impl(c, u, v, pk, gz, divg2, dt, ptop, 0., nullptr, nullptr, delp, gz_scale);
}
extern "C" int fv3_grad1_p_update(fv3_ctx *c, const double *divg2, double *u, double *v, const double *pk, const double *gz, double dt,
                                  double ptk, double beta, double *du, double *dv) {
  if (!du || !dv) return fail("fv3_grad1_p_update: du, dv (U / V x npz, zero before the first call) are required");
  return one_grad_p_impl(c, u, v, pk, gz, divg2, dt, ptk, beta, du, dv);
}
static int one_grad_p_impl(fv3_ctx *c, double *u, double *v, const double *pk, const double *gz, const double *divg2, double dt,
                           double ptk, double beta, double *du, double *dv, const double *delp, double gz_scale) {
  if (!c || !c->grid_ready) return fail("fv3_one_grad_p: context has no grid");
  if (!u || !v || !pk || !gz) return fail("fv3_one_grad_p: null field");
  if (need_scratch(c, delp ? 3 : 2)) return 1;
  const Grid &g = c->g;
  const int km = g.npz;
  constexpr int TI = 32, TJ = 16;
  {
    A2BCorners<TI, TJ> kf;
    kf.g = g;
    kf.in[0] = pk; kf.in[1] = gz; kf.in[2] = kf.in[3] = nullptr;
    kf.out[0] = c->scratch[0]; kf.out[1] = c->scratch[1]; kf.out[2] = kf.out[3] = nullptr;
    kf.nlev[0] = kf.nlev[1] = km + 1;
    kf.nlev[2] = kf.nlev[3] = 0;
    kf.nf = 2;
    if (delp) {               // hydrostatic = .false. (:1996-1997): the layer weights are a2b_ord4 of delp
      kf.in[2] = delp; kf.out[2] = c->scratch[2]; kf.nlev[2] = km; kf.nf = 3;
    }
    for (int f = 0; f < 4; f++) { kf.scale[f] = 1.0; kf.top[f] = 0.; }
    kf.scale[1] = gz_scale;   // gz = zh * grav (:982-989) formed while the tile is staged, as in nh_p_grad
    kf.top[0] = ptk;          // pk(i,j,1) = top_value (:1950-1955): ptk, or ptop where pk is the full pressure
    kf.override_mask = 1;
    RT((run_a2b<TI, TJ>(c, kf, km + 1)));
  }
  {
    OneGradPHydro kf{g, dt, c->scratch[0], c->scratch[1], divg2, u, v};
    kf.beta = beta; kf.du = du; kf.dv = dv;
    kf.dpc = delp ? c->scratch[2] : nullptr;
    Dim3 grid;
    grid.x = (unsigned)(((g.nx + 1) * (g.ny + 1) + OneGradPHydro::CH - 1) / OneGradPHydro::CH);
    grid.y = 1;
    grid.z = (unsigned)km;
    RT(launch_p(c, "one_grad_p", grid, 0, kf));
  }
  return 0;
}

extern "C" int fv3_copy_a_to_cc(fv3_ctx *c, const double *src, double *dst, int nk) {
  if (!c || !c->grid_ready || !src || !dst) return fail("fv3_copy_a_to_cc: bad context/arguments");
  const Grid &g = c->g;
  CopyAtoCC kf{g, src, dst};
  Dim3 grid;
  grid.x = (unsigned)((g.nx * g.ny + CopyAtoCC::CH - 1) / CopyAtoCC::CH);
  grid.y = 1;
  grid.z = (unsigned)nk;
  RT(launch_p(c, "copy_a_to_cc", grid, 0, kf));
  return 0;
}

extern "C" int fv3_pk3_halo(fv3_ctx *c, double ptop, double akap, double *pk3, const double *delp, int use_logp) {
  if (!c || !c->grid_ready) return fail("fv3_pk3_halo: context has no grid");
  Pk3Halo kf{c->g, c->g.npz, use_logp, ptop, akap, delp, pk3};
  Dim3 gr;
  gr.x = (unsigned)((kf.ring() + Pk3Halo::NC - 1) / Pk3Halo::NC);
  gr.y = 1;
  gr.z = 1;
  RT(launch_p(c, "pk3_halo", gr, Pk3Halo::lds_doubles(c->g.npz), kf));
  return 0;
}

extern "C" int fv3_pe_halo(fv3_ctx *c, double ptop, double *pe, const double *delp) {
  if (!c || !c->grid_ready) return fail("fv3_pe_halo: context has no grid");
  PeHalo kf{c->g, c->g.npz, ptop, delp, pe};
  RT(launch_c(c, "pe_halo", col_grid((c->g.nx + 2) * (c->g.ny + 2)), kf));
  return 0;
}

extern "C" int fv3_geopk(fv3_ctx *c, double ptop, double akap, double cp_air, double ptk, double *pe, double *peln,
                         const double *delp, double *pk, double *gz, const double *hs, const double *pt, double *pkz,
                         int CG) {
  if (!c || !c->grid_ready) return fail("fv3_geopk: context has no grid");
  const int e = CG ? 1 : 2;
  const int ncol = (c->g.nx + 2 * e) * (c->g.ny + 2 * e);
  static const int phased_max = [] { const char *v = std::getenv("FV3_MI355X_GEOPK_PHASED"); return v ? std::atoi(v) : 65536; }();
  if (ncol <= phased_max && GeopkPhased::lds_doubles(c->g.npz) * sizeof(double) <= 64 * 1024) {   // small faces: phases over LDS
    GeopkPhased kf{c->g, c->g.npz, CG, ptop, akap, cp_air, ptk, delp, hs, pt, pe, peln, pk, gz, pkz};
    Dim3 gr;
    gr.x = (unsigned)((ncol + GeopkPhased::NC - 1) / GeopkPhased::NC);
    gr.y = 1;
    gr.z = 1;
    RT(launch_p(c, "geopk", gr, GeopkPhased::lds_doubles(c->g.npz), kf));
    return 0;
  }
  Geopk kf{c->g, c->g.npz, CG, ptop, akap, cp_air, ptk, delp, hs, pt, pe, peln, pk, gz, pkz};
  RT(launch_c(c, "geopk", col_grid(ncol), kf));
  return 0;
}

// ================================================================================================
// vertical remap
// ================================================================================================
extern "C" int fv3_set_ak_bk(fv3_ctx *c, const double *ak, const double *bk) {
  if (!c || !ak || !bk) return fail("fv3_set_ak_bk: null argument");
  const int n = c->g.npz + 1;
  if (!c->akbk) RT(rt_malloc((void **)&c->akbk, sizeof(double) * 2 * n));
  RT(rtf_h2d(c->akbk, ak, sizeof(double) * n, c->stream));
  RT(rtf_h2d(c->akbk + n, bk, sizeof(double) * n, c->stream));
  RT(rtf_sync(c->stream));
  c->akbk_ready = true;
  return 0;
}

extern "C" int fv3_set_moist(fv3_ctx *c, const fv3_moist_params *m, double *q_con, double *cappa) {
  if (!c) return fail("fv3_set_moist: null context");
  c->moist_on = m != nullptr;
  if (m) c->moist = *m;
  c->moist_qcon = m ? q_con : nullptr;
  c->moist_cappa = m ? cappa : nullptr;
  return 0;
}

// RemapPar of the energy routines: the scalars of fv3_remap_params + the moist switches of fv3_set_moist
static int energy_par(fv3_ctx *c, const fv3_remap_params *p, RemapPar &rp, const char *who) {
  rp = RemapPar{p->last_step, p->hydrostatic, p->adiabatic, p->nq, p->kord_mt, p->kord_wz, p->kord_tm, p->sphum,
                p->akap, p->ptop, p->rdgas, p->grav, p->cv_air, p->r_vir, p->cp, p->t_min,
                0, 0, 0, 0, 0, 0, 0, 0, 0., 0., 0., nullptr, nullptr, p->fill, c->remap_blocked};
  if (p->sphum < 0 || p->sphum > p->nq) return fail("%s: sphum out of range", who);
  if (c->moist_on && (c->moist.moist_kappa || c->moist.use_cond)) {
    const fv3_moist_params &m = c->moist;
    if (p->hydrostatic) return fail("%s: moist_kappa / use_cond are nonhydrostatic branches", who);
    if (p->sphum < 1 || (m.sphum > 0 && m.sphum != p->sphum)) return fail("%s: moist branches need sphum", who);
    const int idx[5] = {m.liq_wat, m.rainwat, m.ice_wat, m.snowwat, m.graupel};
    for (int n = 0; n < 5; n++)
      if (idx[n] < 0 || idx[n] > p->nq) return fail("%s: water species index out of range", who);
    rp.moist_kappa = m.moist_kappa; rp.use_cond = m.use_cond; rp.nwat = m.nwat;
    rp.liq_wat = m.liq_wat; rp.rainwat = m.rainwat; rp.ice_wat = m.ice_wat; rp.snowwat = m.snowwat; rp.graupel = m.graupel;
    rp.cv_vap = m.cv_vap; rp.c_liq = m.c_liq; rp.c_ice = m.c_ice;
    rp.q_con = c->moist_qcon; rp.cappa = c->moist_cappa;
    if (m.use_cond && !rp.q_con) return fail("%s: use_cond needs q_con (fv3_set_moist)", who);
  }
  return 0;
}

extern "C" int fv3_compute_total_energy(fv3_ctx *c, const fv3_remap_params *p, int moist_phys, const double *u,
                                        const double *v, const double *w, const double *delz, const double *pt,
                                        const double *delp, const double *q, const double *qc, const double *pe,
                                        const double *peln, const double *phis, double *te_2d) {
  if (!c || !c->grid_ready || !p) return fail("fv3_compute_total_energy: bad context/arguments");
  if (!u || !v || !pt || !delp || !phis || !te_2d) return fail("fv3_compute_total_energy: null argument");
  if (p->hydrostatic ? (!pe || !peln) : (!w || !delz)) return fail("fv3_compute_total_energy: null argument of the branch");
  RemapPar rp;
  if (energy_par(c, p, rp, "fv3_compute_total_energy")) return 1;
  const int moist_cvm = !p->hydrostatic && moist_phys && rp.moist_kappa;
  if (moist_cvm && !q) return fail("fv3_compute_total_energy: moist_kappa needs the tracers");
  if (need_scratch(c, 1)) return 1;
  const Grid &g = c->g;
  TotalEnergy kf{g, g.npz, rp, moist_cvm, u, v, w, delz, pt, delp, q, qc, pe, peln, phis, te_2d, c->scratch[0]};
  RT(launch_c(c, "total_energy", col_grid(g.nx * g.ny), kf));
  return 0;
}

extern "C" int fv3_energy_fixer_sums(fv3_ctx *c, const fv3_remap_params *p, int only_sums, const double *u, const double *v,
                                     const double *w, const double *delz, const double *pt, const double *delp,
                                     const double *q, const double *pe, const double *peln, const double *phis,
                                     const double *pkz, const double *pk, const double *te0_2d, double *te_2d,
                                     double *zsum1, double *zsum0) {
  if (!c || !c->grid_ready || !p) return fail("fv3_energy_fixer_sums: bad context/arguments");
  if (!delp || !pkz || !zsum1 || (p->hydrostatic && (!pk || !zsum0))) return fail("fv3_energy_fixer_sums: null argument");
  if (!only_sums) {
    if (!u || !v || !pt || !phis || !te0_2d || !te_2d) return fail("fv3_energy_fixer_sums: null argument");
    if (p->hydrostatic ? (!pe || !peln) : (!w || !delz)) return fail("fv3_energy_fixer_sums: null argument of the branch");
    if (p->sphum > 0 && !q) return fail("fv3_energy_fixer_sums: sphum > 0 needs the tracers");
  }
  RemapPar rp;
  if (energy_par(c, p, rp, "fv3_energy_fixer_sums")) return 1;
  if (c->remap_te_on) { rp.remap_te = 1; rp.hs = c->rte_hs; rp.te = c->rte_te; }   // :655-663: te_2d = sum(te * delp)
  if (need_scratch(c, 1)) return 1;
  const Grid &g = c->g;
  EnergyFixerSums kf{g, g.npz, rp, only_sums, u, v, w, delz, pt, delp, q, pe, peln, phis, pkz, pk, te0_2d, te_2d, zsum1, zsum0, c->scratch[0]};
  RT(launch_c(c, "energy_fixer", col_grid(g.nx * g.ny), kf));
  return 0;
}

extern "C" int fv3_remap_finish(fv3_ctx *c, const fv3_remap_params *p, double dtmp, double *pt, const double *pkz,
                                const double *q) {
  if (!c || !c->grid_ready || !p || !pt || !pkz) return fail("fv3_remap_finish: bad context/arguments");
  if (p->sphum > 0 && !q) return fail("fv3_remap_finish: sphum > 0 needs the tracers");
  RemapPar rp;
  if (energy_par(c, p, rp, "fv3_remap_finish")) return 1;
  const Grid &g = c->g;
  RemapFinish kf{g, g.npz, rp, dtmp, q, pkz, pt};
  Dim3 grid;
  grid.x = (unsigned)((g.nx * g.ny + RemapFinish::CH - 1) / RemapFinish::CH);
  grid.y = 1;
  grid.z = (unsigned)g.npz;
  RT(launch_p(c, "remap_finish", grid, 0, kf));
  return 0;
}

extern "C" int fv3_set_remap_te(fv3_ctx *c, int remap_te, const double *hs, double *te) {
  if (!c) return fail("fv3_set_remap_te: null ctx");
  if (remap_te && (!hs || !te)) return fail("fv3_set_remap_te: remap_te needs hs (A) and te (A x npz)");
  c->remap_te_on = remap_te != 0;
  c->rte_hs = remap_te ? hs : nullptr;
  c->rte_te = remap_te ? te : nullptr;
  return 0;
}

static int remap_two_waves() {   // FV3_MI355X_REMAP_2W=0: the scalars' remap kernel without the two-wavefronts-per-SIMD register budget
  static const int v = [] { const char *e = std::getenv("FV3_MI355X_REMAP_2W"); return e ? std::atoi(e) : 1; }();
  return v;
}
static int remap_probe() {
  const char *e = std::getenv("FV3_MI355X_REMAP_PROBE");
  return e ? std::atoi(e) : 0;
}
extern "C" int fv3_lagrangian_to_eulerian(fv3_ctx *c, const fv3_remap_params *p, const int *kord_tr, double *ps,
                                          double *pe, double *delp, double *pkz, double *pk, double *u, double *v,
                                          double *w, double *delz, double *pt, double *q, double *peln, double *omga,
                                          const double *ws) {
  if (!c || !c->grid_ready || !p) return fail("fv3_lagrangian_to_eulerian: bad context/arguments");
  if (!c->akbk_ready) return fail("fv3_lagrangian_to_eulerian: call fv3_set_ak_bk first");
  if (p->nq < 0 || p->nq > 64) return fail("fv3_lagrangian_to_eulerian: nq out of range");
  if (p->nq > 0 && (!kord_tr || !q)) return fail("fv3_lagrangian_to_eulerian: tracers need q and kord_tr");
  // kord_mt and kord_tr reach the map routines signed, kord_tm and kord_wz as abs() (fv_mapz.F90:359-417, :553)
  if (!kord_supported(p->kord_mt) || !kord_supported(std::abs(p->kord_tm)) || (!p->hydrostatic && !kord_supported(std::abs(p->kord_wz))))
    return fail("fv3_lagrangian_to_eulerian: kord must be <= 15 (8..15: scalar_profile / cs_profile, <= 7: ppm_profile)");
  if (!p->hydrostatic && p->kord_wz < 0)
    return fail("fv3_lagrangian_to_eulerian: kord_wz < 0 (iv=-3) reads an unset array element in the reference; not built");
  for (int n = 0; n < p->nq; n++)
    if (!kord_supported(kord_tr[n])) return fail("fv3_lagrangian_to_eulerian: kord_tr(%d) unsupported", n + 1);
  if (c->g.npz < 5) return fail("fv3_lagrangian_to_eulerian: needs npz > 4 (fv_dynamics.F90:574)");
  const Grid &g = c->g;
  const int km = g.npz;
  if (p->nq > 0) {
    if (!c->kord_tr_dev) RT(rt_malloc((void **)&c->kord_tr_dev, sizeof(int) * 64));
    RT(rtf_h2d(c->kord_tr_dev, kord_tr, sizeof(int) * p->nq, c->stream));
    RT(rtf_sync(c->stream));
  }
  RemapPar rp{p->last_step, p->hydrostatic, p->adiabatic, p->nq, p->kord_mt, p->kord_wz, p->kord_tm, p->sphum,
              p->akap, p->ptop, p->rdgas, p->grav, p->cv_air, p->r_vir, p->cp, p->t_min,
              0, 0, 0, 0, 0, 0, 0, 0, 0., 0., 0., nullptr, nullptr, p->fill, c->remap_blocked};
  const bool moist = c->moist_on && (c->moist.moist_kappa || c->moist.use_cond);
  if (moist) {
    const fv3_moist_params &m = c->moist;
    if (p->hydrostatic) return fail("fv3_lagrangian_to_eulerian: moist_kappa / use_cond are nonhydrostatic branches");
    if (p->sphum < 1 || p->sphum > p->nq || (m.sphum > 0 && m.sphum != p->sphum))
      return fail("fv3_lagrangian_to_eulerian: moist branches need sphum (the same in both parameter sets)");
    const int idx[5] = {m.liq_wat, m.rainwat, m.ice_wat, m.snowwat, m.graupel};
    for (int n = 0; n < 5; n++)
      if (idx[n] < 0 || idx[n] > p->nq) return fail("fv3_lagrangian_to_eulerian: water species index out of range");
    if (m.moist_kappa && (!c->moist_qcon || !c->moist_cappa))
      return fail("fv3_lagrangian_to_eulerian: moist_kappa needs q_con and cappa (fv3_set_moist)");
    rp.moist_kappa = m.moist_kappa; rp.use_cond = m.use_cond; rp.nwat = m.nwat;
    rp.liq_wat = m.liq_wat; rp.rainwat = m.rainwat; rp.ice_wat = m.ice_wat; rp.snowwat = m.snowwat; rp.graupel = m.graupel;
    rp.cv_vap = m.cv_vap; rp.c_liq = m.c_liq; rp.c_ice = m.c_ice;
    rp.q_con = c->moist_qcon; rp.cappa = c->moist_cappa;
  }
  const double *ak = c->akbk, *bk = c->akbk + (km + 1);
  // the column in LDS (remap_fast.h): the spline in the reference's order by hand-over rounds, the rest the slab kernels' code per
  // (column, level); the same bits as the slab kernels below, which keep what it is not built for
  const bool ix32 = (size_t)c->g.nB() * (size_t)(km + 1) < ((size_t)1 << 29);   // 32-bit field indices of the LDS kernels (nh_fast.h ix_t)
  bool fast = c->remap_lds && ix32 && !c->remap_te_on && p->kord_tm < 0 && km <= 127 && km >= 5 && kord_fast(-p->kord_tm) &&
              kord_fast(p->kord_mt) && (p->hydrostatic || kord_fast(p->kord_wz));
  for (int n = 0; n < p->nq && fast; n++) fast = kord_fast(kord_tr[n]);
  // Where they pay: a column of the LDS kernels costs 16 lanes x 8 rows whatever km is (3290 / km ps per cell and field on the C384 /
  // C768 tiles of tools/bench_config5.py), the slab kernels remap the tracers three at a time with one elimination and one search
  // (33 - 36 ps per cell and field when the tracers dominate).  Measured, both ways on one box (config 5's block, 768^2 x 79 with 33
  // tracers): 133 against 101 ms per dt_atmos for the remap, 270 against 233 ms for the step; 384^2 x 127 with 12 tracers: 14.5 against
  // 19.0 ms.  The lines cross at km ~ 97.  FV3_MI355X_REMAP_LDS=2: the LDS kernels wherever they are built.
  // With 5 levels per lane (km <= 79: RemapFast...<..., 5>) no row is idle at L79 and the LDS kernels win there too; the rule is left for
  // 80 <= km <= 96.
  if (fast && c->remap_lds == 1 && km >= 80 && km <= 96 && p->nq >= 8) fast = false;
  if (fast) {
    // levels per lane: 8 (km <= 127) or 5 (km <= 79: 80 rows, no idle lane at L79)
    auto run = [&](auto LV) -> int {
      constexpr int L = decltype(LV)::value;
      {
        const Dim3 gr{(unsigned)((g.nx + kFC - 1) / kFC), (unsigned)g.ny, 1};
        if (p->hydrostatic) {
          using K = RemapFastScalars<true, false, L>;
          RT((remap_two_waves() || L == 5 ? launch_p2<K> : launch_p<K>)(c, "remap_lds_scalars", gr, RLay<L>::Lds, K{g, km, rp, ak, bk, c->kord_tr_dev, pe, ws, ps, delp, pkz, pk, delz, pt, peln, w, q, omga, remap_probe()}));
        } else if (moist) {   // use_cond / moist_kappa (fv3_set_moist): cappa from moist_cv in the temperature transform and in pkz
          using K = RemapFastScalars<false, true, L>;
          RT((launch_p2<K>)(c, "remap_lds_scalars", gr, RLay<L>::Lds, K{g, km, rp, ak, bk, c->kord_tr_dev, pe, ws, ps, delp, pkz, pk, delz, pt, peln, w, q, omga, remap_probe()}));
        } else {
          using K = RemapFastScalars<false, false, L>;
          RT((remap_two_waves() || L == 5 ? launch_p2<K> : launch_p<K>)(c, "remap_lds_scalars", gr, RLay<L>::Lds, K{g, km, rp, ak, bk, c->kord_tr_dev, pe, ws, ps, delp, pkz, pk, delz, pt, peln, w, q, omga, remap_probe()}));
        }
      }
      {
        RemapFastWind<0, L> kf{g, km, p->kord_mt, ak, bk, pe, u};
        RT(launch_p2(c, "remap_lds_winds", Dim3{(unsigned)kf.nblocks_x(), (unsigned)kf.nrows(), 1}, RLay<L>::Lds, kf));
      }
      {
        RemapFastWind<1, L> kf{g, km, p->kord_mt, ak, bk, pe, v};
        RT(launch_p2(c, "remap_lds_winds", Dim3{(unsigned)kf.nblocks_x(), (unsigned)kf.nrows(), 1}, RLay<L>::Lds, kf));
      }
      return 0;
    };
    if (km <= 79) RT(run(std::integral_constant<int, 5>{}));
    else RT(run(std::integral_constant<int, 8>{}));
    RemapPe kf{g, km, ak, bk, pe};
    RT(launch_c(c, "remap_pe", col_grid(g.nx * g.ny), kf));
    return 0;
  }
  // field tasks: T_v, w (nonhydrostatic), u, v, tracer groups -- at most kRemapSets of them per launch, each with its own
  // seven profile slabs; eight coordinate slabs (p, log p, and the face-averaged p of u and of v) in front of them
  // the tracers run in groups of up to c->remap_nt per thread (remap_tracers_col), dealt evenly
  constexpr int kRemapSets = 7, kSetSlabs = RemapFields::kSetSlabs;
  const int ngrp = p->nq > 0 ? (p->nq + c->remap_nt - 1) / c->remap_nt : 0;
  const int ntask = 1 + ngrp + (p->hydrostatic ? 0 : 1) + 2;
  const int nsets = ntask < kRemapSets ? ntask : kRemapSets;
  // a slab holds km+1 levels of every column a task may own (u, v: (nx+1) x (ny+1), in blocks of 64: scr_col)
  const size_t ncol_max = (((size_t)(g.nx + 1) * (g.ny + 1) + 63) / 64) * 64;
  const size_t slab = (g.nA() > ncol_max ? g.nA() : ncol_max) * (size_t)(km + 1);
  const size_t need = slab * (size_t)(8 + kSetSlabs * nsets);
  if (c->remap_scr_n < need) {
    RT(grp_flush_all());   // queued launches of a face group may still hold the old pointer
    if (c->remap_scr) rt_free(c->remap_scr);
    c->remap_scr = nullptr;
    c->remap_scr_n = 0;
    RT(rt_malloc((void **)&c->remap_scr, need * sizeof(double)));
    c->remap_scr_n = need;
  }
  double *co = c->remap_scr, *sets = c->remap_scr + 8 * slab;
  if (c->remap_te_on) {   // fv_mapz.F90:232-286: the energy of every layer from the un-remapped state; u kept for the rows above
    if (!p->hydrostatic && (!w || !delz)) return fail("fv3_lagrangian_to_eulerian: remap_te (nonhydrostatic) needs w and delz");
    if (p->sphum > 0 && !q) return fail("fv3_lagrangian_to_eulerian: remap_te with sphum > 0 needs the tracers");
    double *u_old = cs_scratch(c, 30);
    if (!u_old || need_scratch(c, 1)) return fail("fv3_lagrangian_to_eulerian: out of device memory");
    rp.remap_te = 1; rp.hs = c->rte_hs; rp.te = c->rte_te; rp.u_old = u_old;
    RT(fv3_memcpy_d2d(c, u_old, u, sizeof(double) * g.nU() * (size_t)km));
    RemapTePre kf{g, km, rp, u, v, w, delz, pt, delp, q, pe, pk, peln, pkz, c->scratch[0]};
    RT(launch_c(c, "remap_te_pre", col_grid(g.nx * g.ny), kf));
  }
  {
    RemapCoords kf{g, km, rp, ak, bk, pe, peln, ps, co, co + slab, co + 2 * slab, co + 3 * slab};
    RT(launch_c(c, "remap_coords", col_grid(g.nx * g.ny), kf));
  }
  {
    const int nblk = ((g.nx + 1) * (g.ny + 1) + 255) / 256;  // covers the u and v columns too
    // task order: T_v, w, u, v, tracers.  With moist_kappa the T_v task reads the un-remapped tracers (moist_cv in its
    // source transform), so the tracer tasks go into launches of their own after it.
    const int n_head = 1 + (p->hydrostatic ? 0 : 1) + 2;
    for (int t0 = 0; t0 < ntask;) {
      int nt = ntask - t0 < nsets ? ntask - t0 : nsets;
      if (rp.moist_kappa && t0 < n_head && t0 + nt > n_head) nt = n_head - t0;
      RemapFields kf{g, km, rp, ak, bk, c->kord_tr_dev, delp, pk, delz, peln, pe, ws, w, pt, q, omga, u, v,
                     co, co + slab, co + 2 * slab, co + 3 * slab, co + 4 * slab, co + 5 * slab, co + 6 * slab,
                     co + 7 * slab, sets, slab, t0, nblk, ngrp > 0 ? ngrp : 1};
      Dim3 gr = col_grid(256 * nblk * nt);
      // 248 VGPRs unconstrained (2 wavefronts per SIMD); under the budget of 4 (128 VGPRs, 96 spilled) the k-sequential,
      // latency-bound kernel is 27 % faster (measured: 3 -> 12.6 ms per dt_atmos, 4 -> 11.9, 5 -> 12.5, 6 -> 13.4, 8 -> 15.4,
      // unconstrained 16.4); the Riemann solvers and RemapDelzFinal lose under tighter budgets (spills in their sweeps)
      RT(launch_c<4>(c, "remap_fields", gr, kf));
      t0 += nt;
    }
  }
  {
    ColScr s0{sets, sets + slab, sets + 2 * slab, sets + 3 * slab, sets + 4 * slab, co, co + slab, sets + 5 * slab,
              g.nA(), 0};
    RemapDelzFinal kf{g, km, rp, delp, pkz, pk, delz, pt, peln, q, s0};
    RT(launch_c(c, "remap_delz_final", col_grid(g.nx * g.ny), kf));
  }
  if (rp.remap_te) {   // :576-619 and the conversion of pt (:793-841)
    RemapTePost kf{g, km, rp, ak, bk, u, v, w, delz, delp, q, pe, pk, peln, pt, pkz};
    RT(launch_c(c, "remap_te_post", col_grid(g.nx * g.ny), kf));
  }
  {
    RemapPe kf{g, km, ak, bk, pe};
    RT(launch_c(c, "remap_pe", col_grid(g.nx * g.ny), kf));
  }
  return 0;
}

// ================================================================================================
// tracer_2d
// ================================================================================================
static int need_trc(fv3_ctx *c) {
  const int npz = c->g.npz;
  if (!c->trc_d) RT(rt_malloc((void **)&c->trc_d, sizeof(double) * 3 * npz));   // cmax, frac, trdm per level (cubed deln)
  if (!c->trc_i) RT(rt_malloc((void **)&c->trc_i, sizeof(int) * 2 * npz));      // ksplt, nord_tr per level
  return 0;
}

extern "C" int fv3_tracer_2d_prep(fv3_ctx *c, int q_split, const double *cx, const double *cy, double *xfx,
                                  double *yfx, double *cmax_host) {
  if (!c || !c->grid_ready) return fail("fv3_tracer_2d_prep: context has no grid");
  if (need_trc(c)) return 1;
  const Grid &g = c->g;
  RT(grp_stream_op(c, 1, c->trc_d, nullptr, sizeof(double) * g.npz, 0));
  TracerPrep kf{g, g.npz, q_split, cx, cy, xfx, yfx, c->trc_d};
  const size_t nmax = g.nCX() > g.nCY() ? g.nCX() : g.nCY();
  Dim3 grid;
  grid.x = (unsigned)((nmax + TracerPrep::CH - 1) / TracerPrep::CH);
  grid.y = 1;
  grid.z = (unsigned)g.npz;
  RT(launch_p(c, "tracer_prep", grid, 0, kf));
  if (cmax_host) {
    RT(rtf_d2h(cmax_host, c->trc_d, sizeof(double) * g.npz, c->stream));
    RT(rtf_sync(c->stream));
  }
  return 0;
}

extern "C" int fv3_tracer_2d_scale(fv3_ctx *c, const double *frac_host, double *cx, double *xfx, double *mfx,
                                   double *cy, double *yfx, double *mfy) {
  if (!c || !c->grid_ready || !frac_host) return fail("fv3_tracer_2d_scale: bad context/arguments");
  if (need_trc(c)) return 1;
  const Grid &g = c->g;
  RT(rtf_h2d(c->trc_d + g.npz, frac_host, sizeof(double) * g.npz, c->stream));
  RT(rtf_sync(c->stream));
  TracerScale kf{g, c->trc_d + g.npz, cx, xfx, mfx, cy, yfx, mfy};
  const size_t nmax = g.nCX() > g.nCY() ? g.nCX() : g.nCY();
  Dim3 grid;
  grid.x = (unsigned)((nmax + TracerScale::CH - 1) / TracerScale::CH);
  grid.y = 1;
  grid.z = (unsigned)g.npz;
  RT(launch_p(c, "tracer_scale", grid, 0, kf));
  return 0;
}

static int tracer_step_impl(fv3_ctx *c, int it, int nsplt, const int *ksplt_dev, int nq, int hord, int nord_tr, double trdm,
                            const double *q, double *q_out, const double *dp1, double *dp1_out, const double *mfx, const double *mfy,
                            const double *cx, const double *cy, const double *xfx, const double *yfx, const double *mass);
extern "C" int fv3_tracer_2d_step(fv3_ctx *c, int it, int nsplt, const int *ksplt_host, int nq, int hord, int nord_tr,
                                  double trdm, const double *q, double *q_out, const double *dp1, double *dp1_out,
                                  const double *mfx, const double *mfy, const double *cx, const double *cy,
                                  const double *xfx, const double *yfx) {
  if (!c || !c->grid_ready || !ksplt_host) return fail("fv3_tracer_2d_step: bad context/arguments");
  if (!tp_ord_supported_tr(hord)) return fail("fv3_tracer_2d_step: hord=%d not supported (5,-5,6,7,8,9,10,11,12,13)", hord);
  if (q == q_out || dp1 == dp1_out) return fail("fv3_tracer_2d_step: *_out buffers must not alias the inputs");
  if (trdm > 1.e-4 && nord_tr > 2) return fail("fv3_tracer_2d_step: nord_tr > 2");
  if (need_trc(c)) return 1;
  if (it == 1) {
    RT(rtf_h2d(c->trc_i, ksplt_host, sizeof(int) * c->g.npz, c->stream));
    RT(rtf_sync(c->stream));
  }
  return tracer_step_impl(c, it, nsplt, c->trc_i, nq, hord, nord_tr, trdm, q, q_out, dp1, dp1_out, mfx, mfy, cx, cy, xfx, yfx, nullptr);
}

// ---- inline_q: the tracers advected inside d_sw, every acoustic substep (sw_core.F90:1020-1043) -----------------------------------
// One sub-step of tracer_2d is the same arithmetic (fv_tracer2d.F90:478-520 against sw_core.F90:1021-1043: dp2 / delp, ra_x / ra_y,
// fv_tp_2d with the mass fluxes, the update of q), with d_sw's per-substep Courant numbers, area fluxes and delp fluxes in the
// place of the accumulated ones; deln_flux takes nord_t / damp_t and d_sw's half-updated delp as mass (InlineQMass).
extern "C" int fv3_d_sw_inline_q(fv3_ctx *c, int nq, int hord_tr, int nord_t, double damp_t, const double *q, double *q_out,
                                 const double *delp_old, const double *delp_new, const double *fx, const double *fy,
                                 const double *crx, const double *cry, const double *xfx, const double *yfx) {
  if (!c || !c->grid_ready) return fail("fv3_d_sw_inline_q: context has no grid");
  if (nq < 1 || !q || !q_out || !delp_old || !delp_new || !fx || !fy || !crx || !cry || !xfx || !yfx)
    return fail("fv3_d_sw_inline_q: null argument");
  if (!tp_ord_supported_tr(hord_tr)) return fail("fv3_d_sw_inline_q: hord_tr=%d not supported (5,-5,6,7,8,9,10,11,12,13)", hord_tr);
  if (q == q_out) return fail("fv3_d_sw_inline_q: q_out must not alias q");
  if (damp_t > 1.e-4 && nord_t > 2) return fail("fv3_d_sw_inline_q: nord_t > 2");
  if (need_trc(c)) return 1;
  const Grid &g = c->g;
  if (!c->ones_i) {   // ksplt = 1 on every level, device resident (first call: outside a graph capture)
    std::vector<int> one(g.npz, 1);
    RT(rt_malloc((void **)&c->ones_i, sizeof(int) * g.npz));
    RT(rtf_h2d(c->ones_i, one.data(), sizeof(int) * g.npz, c->stream));
    RT(rtf_sync(c->stream));
  }
  const double *mass = nullptr;
  if (damp_t > 1.e-4) {
    double *m = cs_scratch(c, 29);
    if (!m) return fail("fv3_d_sw_inline_q: out of device memory");
    InlineQMass kf{g, delp_old, delp_new, m};
    Dim3 grid;
    grid.x = (unsigned)((g.nA() + InlineQMass::CH - 1) / InlineQMass::CH);
    grid.y = 1;
    grid.z = (unsigned)g.npz;
    RT(launch_p(c, "inline_q_mass", grid, 0, kf));
    mass = m;
  }
  return tracer_step_impl(c, 1, 1, c->ones_i, nq, hord_tr, nord_t, damp_t, q, q_out, delp_old, nullptr, fx, fy, crx, cry, xfx, yfx, mass);
}

extern "C" int fv3_flux_accum(fv3_ctx *c, double *mfx, double *mfy, const double *fx, const double *fy) {
  if (!c || !c->grid_ready || !mfx || !mfy || !fx || !fy) return fail("fv3_flux_accum: bad context/arguments");
  const Grid &g = c->g;
  FluxAccum kf{g, mfx, mfy, fx, fy};
  const size_t nmax = g.nFX() > g.nFY() ? g.nFX() : g.nFY();
  Dim3 grid;
  grid.x = (unsigned)((nmax + FluxAccum::CH - 1) / FluxAccum::CH);
  grid.y = 1;
  grid.z = (unsigned)g.npz;
  RT(launch_p(c, "flux_accum", grid, 0, kf));
  return 0;
}

extern "C" int fv3_fill2d_mass(fv3_ctx *c, int nk, const double *q, const double *delp, double *qt) {
  if (!c || !c->grid_ready || nk < 1 || !q || !delp || !qt) return fail("fv3_fill2d_mass: bad context/arguments");
  const Grid &g = c->g;
  Fill2dMass kf{g, q, delp, qt};
  Dim3 grid;
  grid.x = (unsigned)(((size_t)g.nx * g.ny + Fill2dMass::CH - 1) / Fill2dMass::CH);
  grid.y = 1;
  grid.z = (unsigned)nk;
  RT(launch_p(c, "fill2d_mass", grid, 0, kf));
  return 0;
}
extern "C" int fv3_fill2d_apply(fv3_ctx *c, int nk, const double *qt, const double *delp, double *q) {
  if (!c || !c->grid_ready || nk < 1 || !q || !delp || !qt) return fail("fv3_fill2d_apply: bad context/arguments");
  const Grid &g = c->g;
  Fill2dApply kf{g, qt, delp, q};
  Dim3 grid;
  grid.x = (unsigned)(((size_t)g.nx * g.ny + Fill2dApply::CH - 1) / Fill2dApply::CH);
  grid.y = 1;
  grid.z = (unsigned)nk;
  RT(launch_p(c, "fill2d_apply", grid, 0, kf));
  return 0;
}

static int tracer_step_impl(fv3_ctx *c, int it, int nsplt, const int *ksplt_dev, int nq, int hord, int nord_tr, double trdm,
                            const double *q, double *q_out, const double *dp1, double *dp1_out, const double *mfx, const double *mfy,
                            const double *cx, const double *cy, const double *xfx, const double *yfx, const double *mass) {
  const Grid &g = c->g;
  auto march_step = [&]() -> int {
    const int trc_nt = c->trc_nt;   // tracers per wavefront (1: one (tracer, level) per wavefront, TracerMarch)
    if (trc_nt > 1 && nq > 1) {
      auto go = [&](auto H, auto NTc) -> int {
        constexpr int NT = decltype(NTc)::value;
        const int ngrp = (nq + NT - 1) / NT;
        MarchDims md = make_march_dims(g, seg_rows(c, c->march_tj_fused, g.npz * ngrp));
        const int nwt = md.nwaves(g.npz * ngrp);
        TracerMarchFused<decltype(H)::value, NT> kf{g, md, g.npz, nq, it, nsplt, ngrp, ksplt_dev, q, dp1, mfx, mfy, cx,
                                                    cy, xfx, yfx, q_out, dp1_out};
        return launch_w(c, "tracer_step", nwt, kf);
      };
      return dispatch_hord_tr(hord, [&](auto H) {
        const int nt = nq < trc_nt ? nq : trc_nt;
        if (nt == 4) return go(H, std::integral_constant<int, 4>{});
        if (nt == 3) return go(H, std::integral_constant<int, 3>{});
        return go(H, std::integral_constant<int, 2>{});
      });
    }
    MarchDims md = make_march_dims(g, seg_rows(c, c->march_tj, g.npz));
    const int nwt = md.nwaves(g.npz * nq);
    return dispatch_hord_tr(hord, [&](auto H) {
      TracerMarch<decltype(H)::value> kf{g, md, g.npz, nq, it, nsplt, ksplt_dev, q, dp1, mfx, mfy, cx, cy, xfx, yfx,
                                         q_out, dp1_out};
      return launch_w(c, "tracer_step", nwt, kf);
    });
  };
  if (is_cubed(c)) {
    double *fx = cs_scratch(c, 8), *fy = cs_scratch(c, 9);
    if (!fx || !fy) return fail("fv3_tracer_2d_step: out of device memory");
    const bool damp = it == 1 && trdm > 1.e-4;      // fv_tracer2d.F90:497-505: deln_flux inside fv_tp_2d, mass = dp1
    int *nord_dev = nullptr;
    double *coef_dev = nullptr;
    if (damp) {  // the order and the coefficient as per-level arrays (cubed_damp.h takes them per level): trc_i / trc_d slots
      std::vector<int> ni(g.npz, nord_tr);
      std::vector<double> cd(g.npz, trdm);
      nord_dev = c->trc_i + g.npz;
      coef_dev = c->trc_d + 2 * g.npz;
      RT(rtf_h2d(nord_dev, ni.data(), sizeof(int) * g.npz, c->stream));
      RT(rtf_h2d(coef_dev, cd.data(), sizeof(double) * g.npz, c->stream));
      RT(rtf_sync(c->stream));
    }
    // Hybrid (see dsw_cubed): the marching kernels over the whole face (q_out, dp1_out are not inputs), then the cubed fv_tp_2d
    // passes and the update on the frame along the face edges, tracer by tracer
    const int wo = c->cubed_frame, wm = wo + c->cubed_reach;
    const bool hyb = !damp && c->use_march && wo > 0 && g.npx == g.npy && g.npx - 1 >= 2 * wm + 8;
    if (hyb) RT(march_step());
    const PassRegion rm{hyb ? wm : 0, nullptr, g.npz}, ro{hyb ? wo : 0, nullptr, g.npz};
    const size_t nq3 = (size_t)g.npz * g.nA();
    for (int iq = 0; iq < nq; iq++) {
      if (frame_fused_on()) {
        const TpfField fl[3] = {TpfField{q + iq * nq3, fx, fy, hord}, TpfField{}, TpfField{}};
        RT(tp2d_frame_fused(c, fl, 1, cx, cy, xfx, yfx, wo + 1, nullptr, g.npz, "trc_tp", !hyb, mfx, mfy));
      } else if (tp2d_cubed(c, g.npz, q + iq * nq3, cx, cy, hord, fx, fy, xfx, yfx, nullptr, nullptr, mfx, mfy, "trc_tp", &rm)) {
        return 1;
      }
      if (damp) {
        DelnCubedState d;
        d.g = g; d.q = q + iq * nq3; d.mass = mass ? mass : dp1; d.fx = fx; d.fy = fy; d.nord = nord_dev; d.coef = coef_dev; d.thresh = 1.E-4;
        d.corner_area = 0;
        d.d2 = cs_scratch(c, 4); d.fx2 = cs_scratch(c, 5); d.fy2 = cs_scratch(c, 6);
        if (!d.d2 || !d.fx2 || !d.fy2) return fail("fv3_tracer_2d_step: out of device memory");
        RT(launch_pass(c, "trc_deln", g.isd, g.ied, g.jsd, g.jed, rm, DelnCubedL1{d}));
        RT(launch_pass(c, "trc_deln", g.isd, g.ied + 1, g.jsd, g.jed + 1, rm, DelnCubedL24{d, 1, 0}));
        for (int n = 1; n <= nord_tr; n++) {
          RT(launch_pass(c, "trc_deln", g.isd, g.ied, g.jsd, g.jed, rm, DelnCubedL3{d, n}));
          RT(launch_pass(c, "trc_deln", g.isd, g.ied + 1, g.jsd, g.jed + 1, rm, DelnCubedL24{d, 0, n}));
        }
        RT(launch_pass(c, "trc_deln", g.is, g.ie + 1, g.js, g.je + 1, rm, DelnCubedL5{d}));
      }
      TracerCubedFinal kf{g, it, nsplt, iq == nq - 1, ksplt_dev, q + iq * nq3, dp1, fx, fy, mfx, mfy, q_out + iq * nq3, dp1_out};
      RT(launch_pass(c, "trc_fin", g.is, g.ie, g.js, g.je, ro, kf));
    }
    return 0;
  }
  if (c->use_march && !(it == 1 && trdm > 1.e-4)) return march_step();
  constexpr int TI = FV3_DSW_TI, TJ = FV3_DSW_TJ;
  TracerStep<TI, TJ> kf{g, g.npz, nq, it, nsplt, hord, nord_tr, trdm, ksplt_dev, q, dp1, mfx, mfy, cx, cy, xfx, yfx,
                        q_out, dp1_out, mass};
  Dim3 grid;
  grid.x = (unsigned)((g.nx + TI - 1) / TI);
  grid.y = (unsigned)((g.ny + TJ - 1) / TJ);
  grid.z = (unsigned)g.npz;
  RT(launch_p(c, "tracer_step", grid, TracerStep<TI, TJ>::lds_doubles, kf));
  return 0;
}
