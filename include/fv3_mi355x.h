/*
 * fv3_mi355x.h -- C ABI of the MI355X-native FV3 dyn_core hot path (libfv3_mi355x.so).
 *
 * The reference (NOAA-GFDL/GFDL_atmos_cubed_sphere, release 202411) has no FFI: its boundary is
 * the set of Fortran module procedures that dyn_core/fv_dynamics call.
 * Each entry point below stands behind one of those call sites; the comment on each names the
 * reference interface it replaces (file:line under the reference tree).  INTEGRATION.md shows
 * the ISO_C_BINDING interface block a maintainer would add on the Fortran side.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only.  Every function returns 0 on success, non-zero
 *     on error (the reference has no status codes: it calls mpp_error(FATAL); a Fortran shim
 *     should `error stop` on non-zero).  fv3_last_error() returns a message.
 *   - Field arguments are DEVICE pointers to arrays in the reference's own layout (Fortran
 *     column-major, i fastest, then j, then k) with the exact bounds of model/fv_arrays.F90:
 *     1521-1563 and model/dyn_core.F90:256-283; the "kind" tags below give the horizontal
 *     extent of one k-slab:
 *        A  (isd:ied,   jsd:jed)      U  (isd:ied,   jsd:jed+1)   V  (isd:ied+1, jsd:jed)
 *        B  (isd:ied+1, jsd:jed+1)    CX (is:ie+1,   jsd:jed)     CY (isd:ied,   js:je+1)
 *        FX (is:ie+1,   js:je)        FY (is:ie,     js:je+1)     CC (is:ie,     js:je)
 *     All entry points process every level k = 1..npz in one call (the reference calls the 2-D
 *     routines from an OpenMP k loop, model/dyn_core.F90:436-447,658-812).
 *   - Fields the reference updates in place but whose halo other cells still have to read in
 *     the same sweep (delp, pt, w, u, v, q_con in d_sw) have separate *_out arguments that must
 *     not alias the inputs; only the compute domain of an *_out array is written, its halo is
 *     the caller's to refresh (exactly where the reference calls start_group_halo_update,
 *     model/dyn_core.F90:823-825,1169).
 *   - Work is enqueued on the context's HIP stream (fv3_set_stream); nothing synchronises.
 *   - Supported branch sets: grid_type = 4 (doubly periodic / Cartesian branches of the reference) with array-valued
 *     metric terms, and grid_type < 3 (a whole face of the cubed sphere per context: face edges and corners, the damping,
 *     heating, dissipation-estimate and condensate branches included); no nesting, no regional BCs.  A branch that
 *     is not built returns non-zero with the reason in fv3_last_error() -- never a silent fallback.
 */
#ifndef FV3_MI355X_H
#define FV3_MI355X_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fv3_ctx fv3_ctx;

/* fv_grid_bounds_type (model/fv_arrays.F90:1192-1200) + the flagstruct members the kernels read
 * (model/sw_core.F90:126-127,590-591,624,964,1250). */
typedef struct fv3_domain {
  int is, ie, js, je; /* compute domain (global indices of this rank's block) */
  int ng;             /* halo width, 3 (tools/fv_mp_mod.F90:61) */
  int npx, npy, npz;  /* global corner counts and number of levels */
  int grid_type;      /* 4 = doubly periodic (Cartesian), 0..2 = a face of the cubed sphere */
  int do_diss_est, prevent_diss_cooling, stretched_grid;
  double lim_fac;
} fv3_domain;

/* gridstruct members (model/fv_arrays.F90:75-205; shapes :1749-1881).  HOST pointers; copied to
 * the device once by fv3_grid_upload. */
typedef struct fv3_grid_host {
  double da_min, da_min_c;
  const double *area, *rarea, *dxa, *dya, *rdxa, *rdya, *cosa_s, *rsin2, *f0;       /* A */
  const double *dx, *rdx, *dyc, *rdyc, *cosa_v, *sina_v, *rsin_v, *divg_u, *del6_u; /* U */
  const double *dy, *rdy, *dxc, *rdxc, *cosa_u, *sina_u, *rsin_u, *divg_v, *del6_v; /* V */
  const double *rarea_c, *fC, *cosa, *sina;                                          /* B */
  const double *sin_sg, *cos_sg;                                                     /* A x 9 */
} fv3_grid_host;

/* The extra gridstruct members a cubed-sphere face needs (grid_type < 3; one context = one whole face, is = js = 1,
 * ie = je = npx - 1): the A -> B interpolation weights on the four face edges (edge_w / edge_e (npy), edge_s / edge_n (npx),
 * model/fv_grid_utils.F90:1121-1230), rsina (is:ie+1, js:je+1) and the extrap_corner factors x1 / (x2 - x1) of a2b_ord4
 * (model/a2b_edge.F90:83-112, :452-462; corners sw, se, ne, nw x the three centre pairs in the reference's order).
 * HOST pointers, copied once; call after fv3_grid_upload. */
typedef struct fv3_grid_cubed {
  const double *edge_w, *edge_e, *edge_s, *edge_n;
  const double *rsina;
  double corner_f[12];
  /* cubed_to_latlon (init_cubed_to_latlon, fv_grid_utils.F90:2255-2315): a11, a12, a21, a22 on the A layout (isd:ied, jsd:jed);
   * NULL = fv3_c2l is not used on this face */
  const double *a11, *a12, *a21, *a22;
  /* unit vectors of adv_pe (dyn_core.F90:1529-1632), component-last planes: ec1, ec2 (get_center_vect, fv_grid_utils.F90:1738)
   * A layout x 3; en1 (is:ie, js:je+1) x 3 and en2 (is:ie+1, js:je) x 3 (:632-643).  NULL = fv3_adv_pe is not used */
  const double *ec1, *ec2, *en1, *en2;
} fv3_grid_cubed;

const char *fv3_last_error(void);
int fv3_create(const fv3_domain *dom, fv3_ctx **out);
int fv3_destroy(fv3_ctx *ctx);
/* stream: a hipStream_t (NULL = default stream). */
int fv3_set_stream(fv3_ctx *ctx, void *stream);
int fv3_grid_upload(fv3_ctx *ctx, const fv3_grid_host *g);
/* The faces (tiles) one rank holds, launched together.  The reference runs the tiles of a PE one after the other through the same
 * code (tools/fv_mp_mod.F90:276-392 domain_decomp: `tile` / `ntiles_g` per PE; model/dyn_core.F90 is called once per tile the PE
 * owns); six contexts on one MI355X would likewise issue every kernel six times, and at C96 - C384 the pass / frame kernels of a face
 * fill a fraction of the chip.  Members of a group QUEUE their launches; when every member has issued the same kernel with the same
 * grid, ONE launch runs all of them (the face is the slowest grid index, the functors travel by value in one kernel-argument block).
 * The results are those of the separate launches bit for bit (the same code on the same data).  The caller issues each call for all
 * members in turn (any order of members, the same order of calls); everything that reads results or touches another stream -- memcpy
 * to or from the host, fv3_sync, halo exchanges, fv3_gather_run -- runs what is queued first, face by face where the members are not
 * at the same kernel.  All members launch on the first member's stream (fv3_set_stream on any member moves all of them).
 * fv3_group_stats: launches since the last call that ran all members at once / that ran alone. */
typedef struct fv3_group fv3_group;
int fv3_group_create(fv3_ctx *const *members, int n, fv3_group **out);   /* 1 <= n <= 6 */
int fv3_group_flush(fv3_group *grp);
int fv3_group_stats(fv3_group *grp, long *merged, long *single);
int fv3_group_destroy(fv3_group *grp);
int fv3_grid_upload_cubed(fv3_ctx *ctx, const fv3_grid_cubed *g);
/* Geometry mode fv3_grid_upload found in the metric arrays (or -1 without a grid): 0 = general (every metric row is
 * read); 1 = orthogonal: cosa_s, cosa_u, cosa_v = 0 and rsin2, sina_u, sina_v, rsin_u, rsin_v, sin_sg(:,:,1:4) = 1
 * in every element, what fv_grid_utils.F90:427 / fv_grid_tools.F90:1202-1221 set for grid_type >= 3 -- the kernels
 * then skip those rows (x*1 and x - y*0 are exact, so the results do not change); 2 = orthogonal and every length /
 * area array spatially constant (Cartesian doubly periodic, dx_const / dy_const): they travel as scalars.
 * FV3_MI355X_GEOM=<n> in the environment caps the mode (0 forces the general kernels). */
int fv3_grid_geom(const fv3_ctx *ctx);

/* Device-memory helpers for callers that do not own a device allocator (a Fortran host). */
int fv3_malloc(void **dptr, size_t bytes);
int fv3_free(void *dptr);
int fv3_memcpy_h2d(fv3_ctx *ctx, void *dst, const void *src, size_t bytes);
int fv3_memcpy_d2h(fv3_ctx *ctx, void *dst, const void *src, size_t bytes);
int fv3_memset(fv3_ctx *ctx, void *dst, int value, size_t bytes);
int fv3_memcpy_d2d(fv3_ctx *ctx, void *dst, const void *src, size_t bytes);
int fv3_sync(fv3_ctx *ctx);

/* Host-address field registry (SURVEY 8(b) "device-resident registry").  The reference's entry points (dyn_core, fv_dynamics) are handed
 * HOST arrays the caller owns (fv_arrays.F90:1417), so a wrapper with their argument list copies every array in and out on every call
 * unless it knows which copy is current.  fv3_registry_put / _get are the wrapper's h2d / d2h through an entry keyed by the host address:
 *   eager (default): both copy every time -- the caller may have touched anything between the calls;
 *   lazy (fv3_registry_mode(ctx, 1)): put copies only when the device mirror is not current (first use, or the caller declared a write
 *   with fv3_registry_host_touched(ctx, host); host = NULL: every array), get only marks the host copy stale; the caller brings an array
 *   it wants to read with fv3_registry_fetch(ctx, host) (NULL: every stale array).
 * fv3_registry_stats: {h2d copies, h2d skipped, d2h copies, d2h deferred}.  The Fortran wrappers go through it
 * (fortran/fv3_dyn_core_mod.F90: fv3_dyn_core_registry, fv3_host_touched, fv3_host_fetch).
 * An entry is keyed by the host ADDRESS and does not know the array's lifetime: lazy mode needs contiguous actual arguments that stay
 * where they are between the calls (a non-contiguous actual reaches the wrapper as a compiler temporary, i.e. as the address of a
 * copy); an array that is deallocated or rebound is taken out with fv3_registry_forget(ctx, host, discard) (host = NULL: every entry;
 * discard = 0 fetches what the host copy lacks first). */
int fv3_registry_mode(fv3_ctx *ctx, int lazy);
int fv3_registry_put(fv3_ctx *ctx, void *dev, const void *host, size_t bytes);
int fv3_registry_get(fv3_ctx *ctx, void *host, const void *dev, size_t bytes);
int fv3_registry_host_touched(fv3_ctx *ctx, const void *host);
int fv3_registry_fetch(fv3_ctx *ctx, void *host);
int fv3_registry_forget(fv3_ctx *ctx, void *host, int discard);
int fv3_registry_stats(fv3_ctx *ctx, long long *out4);

/* fv_tp_2d -- model/tp_core.F90:85-87 (called from sw_core.F90:919,983,993,1014,1498,
 * nh_utils.F90:279,289, fv_tracer2d.F90:504,509).  nk slabs.  q: A; crx,xfx: CX; cry,yfx: CY;
 * ra_x: (is:ie, jsd:jed); ra_y: (isd:ied, js:je); fx,mfx: FX; fy,mfy: FY; mass: A.
 * Optional arguments are NULL / nord < 0 when absent.  hord: 5, -5, 6, 7, 8, 9, 10, 11, 12, 13 (xppm / yppm, :365-707;
 * d_sw and update_dz_d take 5, -5, 6, 8, 10). */
int fv3_fv_tp_2d(fv3_ctx *ctx, int nk, const double *q, const double *crx, const double *cry, int hord,
                 double *fx, double *fy, const double *xfx, const double *yfx, const double *ra_x,
                 const double *ra_y, const double *mfx, const double *mfy, const double *mass, int nord,
                 double damp_c);

/* xppm / yppm on ONE line -- model/tp_core.F90:324-712 / :715-1152 (private to tp_core_mod: the reference reaches them through
 * fv_tp_2d only).  A unit-test surface like fv3_fv_tp_2d: the reference-held vectors of the 1-D operator (tests/golden/ppm1d_golden.npz,
 * from the reference's docs/examples/tp_core.ipynb) go through it, iord 10 among them -- inside fv_tp_2d the inner sweep of hord 10 is
 * ord 8 (:136-141), so its vectors cannot pass through that entry unchanged.
 * h: the line with 3 halo cells on either side (n + 6), c: n + 1 Courant numbers, flux: n + 1 face values (device pointers).
 * which: 0 the operator of the LDS-tile kernels (every hord of fv3_fv_tp_2d), 1 / 2 the operators of the marching kernels along
 * the lanes (n <= 58) / through the register window (iord 5, -5, 6, 8, 10). */
int fv3_ppm_line(fv3_ctx *ctx, int iord, int which, const double *h, const double *c, double *flux, int n);

/* c_sw -- model/sw_core.F90:79-81, the k loop of model/dyn_core.F90:436-447.
 * in : delp, pt, w (A; w NULL if hydrostatic), u (U), v (V)
 * out: delpc, ptc, wc (A, valid is-1:ie+1 x js-1:je+1); uc (V), vc (U): C-grid winds advanced half a
 *      step on is:ie+1 x js:je / is:ie x js:je+1 and interpolated values on the one-cell ring around
 *      that; ua, va (A, valid is-1:ie+1); ut, vt (A, time-scaled contravariant fluxes, valid
 *      is-1:ie+2 x js-1:je+1 / is-1:ie+1 x js-1:je+2); divg_d (B, valid is-1:ie+2 x js-1:je+2; only
 *      if nord > 0). */
int fv3_c_sw(fv3_ctx *ctx, double *delpc, const double *delp, double *ptc, const double *pt, const double *u,
             const double *v, const double *w, double *uc, double *vc, double *ua, double *va, double *wc,
             double *ut, double *vt, double *divg_d, int nord, double dt2, int hydrostatic, int dord4);

/* Scalars of d_sw's argument list (model/sw_core.F90:494-500) that do not vary with k ... */
typedef struct fv3_dsw_params {
  double dt;
  int hord_tr, hord_mt, hord_vt, hord_tm, hord_dp;
  double dddmp, d4_bg, kgb;
  int hydrostatic, use_cond;
} fv3_dsw_params;

/* ... and the per-level ones dyn_core computes inside its k loop (model/dyn_core.F90:666-733):
 * HOST arrays of length npz. */
typedef struct fv3_dsw_levels {
  const int *nord_k, *nord_v, *nord_w, *nord_t;
  const double *d2_divg, *damp_vt, *damp_w, *damp_t, *d_con_k;
} fv3_dsw_levels;

/* Copies the per-level coefficients to the device; they stay valid for later fv3_d_sw calls. */
int fv3_dsw_levels_upload(fv3_ctx *ctx, const fv3_dsw_levels *lv);

/* d_sw -- model/sw_core.F90:494-500, the k loop of model/dyn_core.F90:658-812 (inline_q=.false.).
 * in    : delp, pt, w, q_con (A), u (U), v (V), uc (V), vc (U), ua, va (A; read only when a level
 *         has nord_k = 0), divg_d (B; read when nord_k > 0)
 * inout : mfx (FX), mfy (FY), cx (CX), cy (CY)  -- the flux capacitors, accumulated
 * out   : crx, xfx (CX), cry, yfx (CY); delp_out, pt_out, w_out, q_con_out (A, compute domain);
 *         u_out (U, is:ie x js:je+1), v_out (V, is:ie+1 x js:je) -- still scaled by dx, dy exactly as
 *         the reference leaves them (sw_core.F90:1233,1502); heat_s, diss_e (CC, the per-call 2-D
 *         heat_source / diss_est of sw_core.F90:521-522, stacked in k; either may be NULL when the caller
 *         does not read it -- dyn_core.F90:798-812 does when d_con > 1e-5 or do_diss_est); delpc (A, the
 *         saved divergence on is:ie+1 x js:je+1; may be NULL).
 * The reference's clobbering of uc, vc, divg_d as scratch (sw_core.F90:1394-1408) is not reproduced:
 * those arrays are left unchanged. */
int fv3_d_sw(fv3_ctx *ctx, const fv3_dsw_params *p, double *delpc, const double *delp, const double *pt,
             const double *u, const double *v, const double *w, const double *uc, const double *vc,
             const double *ua, const double *va, const double *divg_d, double *mfx, double *mfy, double *cx,
             double *cy, double *crx, double *cry, double *xfx, double *yfx, const double *q_con,
             double *delp_out, double *pt_out, double *u_out, double *v_out, double *w_out, double *q_con_out,
             double *heat_s, double *diss_e);
/* The same routine in two calls, for overlapping the halo exchange of uc, vc, divg_d (dyn_core.F90:451, :565-578 --
 * where the reference overlaps its start/complete_group_halo_update pairs with compute) with the part of d_sw that
 * does not read those halos: fv3_d_sw_interior runs the interior strips / segments of the fused transport kernel and
 * may be called before the halos are complete; fv3_d_sw_rest (same arguments) does everything else afterwards.
 * interior + rest == fv3_d_sw bit for bit; when the fused kernel is not the active path interior is a no-op. */
int fv3_d_sw_interior(fv3_ctx *ctx, const fv3_dsw_params *p, double *delpc, const double *delp, const double *pt,
             const double *u, const double *v, const double *w, const double *uc, const double *vc,
             const double *ua, const double *va, const double *divg_d, double *mfx, double *mfy, double *cx,
             double *cy, double *crx, double *cry, double *xfx, double *yfx, const double *q_con,
             double *delp_out, double *pt_out, double *u_out, double *v_out, double *w_out, double *q_con_out,
             double *heat_s, double *diss_e);
int fv3_d_sw_rest(fv3_ctx *ctx, const fv3_dsw_params *p, double *delpc, const double *delp, const double *pt,
             const double *u, const double *v, const double *w, const double *uc, const double *vc,
             const double *ua, const double *va, const double *divg_d, double *mfx, double *mfy, double *cx,
             double *cy, double *crx, double *cry, double *xfx, double *yfx, const double *q_con,
             double *delp_out, double *pt_out, double *u_out, double *v_out, double *w_out, double *q_con_out,
             double *heat_s, double *diss_e);

/* Periodic halo fill of one doubly periodic tile owned by a single rank (the two periodic contacts
 * of tools/fv_mp_mod.F90:473-483 when layout = 1x1): the single-GPU replacement of
 * start/complete_group_halo_update (tools/fv_mp_mod.F90:646-876).  kind: 0=A 1=U 2=V 3=B. */
int fv3_halo_fill_periodic(fv3_ctx *ctx, double *field, int kind, int nk);

/* Multi-rank halo exchange of a block-decomposed domain: pack / unpack kernels that replace the buffer side of
 * mpp_update_domains (tools/fv_mp_mod.F90:646-876 group updates; FMS mpp_domains underneath).  Message d
 * (0..7: the neighbour offsets (di,dj), dj = -1,0,1 outer, di = -1,0,1 inner, (0,0) skipped) carries, for every
 * field of the group, the 3-wide interior strip adjacent to the d-side boundary; it is received into the halo of
 * the opposite side (-d) of the neighbour at offset d.  fv3_halo_message_elems gives the length of each message;
 * pack fills the 8 send buffers from the fields, unpack scatters the 8 received buffers into the halos (message
 * d lands in the (-d)-side halo).  The transfers themselves are the caller's (RCCL send/recv, see halo.py).
 * kind: 0 = A (CENTER), 1 = U (y-staggered), 2 = V (x-staggered), 3 = B (CORNER); the staggered edge row /
 * column ie+1 / je+1 belongs to its owner and is never overwritten. */
#define FV3_HALO_MAX_FIELDS 8
typedef struct fv3_halo_field {
  double *field;
  int kind, nk;
} fv3_halo_field;
int fv3_halo_message_elems(fv3_ctx *ctx, int nfields, const fv3_halo_field *fields, size_t elems[8]);
/* One rank of a doubly periodic domain: the group's halos filled straight from its own opposite edges, one launch -- what
 * fv3_halo_pack + fv3_halo_unpack do through the eight message buffers when every neighbour is the rank itself
 * (mpp_update_domains on a 1 x 1 periodic layout, tools/fv_mp_mod.F90:473-483). */
int fv3_halo_periodic_group(fv3_ctx *ctx, int nfields, const fv3_halo_field *fields);
int fv3_halo_pack(fv3_ctx *ctx, int nfields, const fv3_halo_field *fields, double *const sendbuf[8]);
int fv3_halo_unpack(fv3_ctx *ctx, int nfields, const fv3_halo_field *fields, const double *const recvbuf[8]);

/* The transfers behind the C ABI: RCCL peer send / recv on a second HIP stream owned by the context -- what
 * start_group_halo_update / complete_group_halo_update (tools/fv_mp_mod.F90:646-876) do over FMS / MPI, for hosts (the
 * Fortran dyn_core) that have no RCCL binding of their own.
 *   fv3_comm_get_unique_id  on rank 0; the host distributes the 128 bytes with its own means (MPI_Bcast)
 *   fv3_comm_init           every rank: ncclCommInitRank; librccl.so is loaded at run time, there is no link dependency
 *   fv3_halo_start          pack kernel, then one grouped send + recv per neighbour offset d on the communication stream
 *                           (to[d] / from[d]: the ranks at offsets d / -d; a rank may be its own neighbour); kernels launched
 *                           on the context's stream before fv3_halo_complete overlap the transfers (dyn_core.F90:565-578)
 *   fv3_halo_complete       the context's stream waits for the transfers, unpack kernel
 *   fv3_allreduce_max       mp_reduce_max (tools/fv_mp_mod.F90:1683; tracer_2d's cmax, fv_tracer2d.F90:405): n HOST doubles, in place */
#define FV3_COMM_ID_BYTES 128
int fv3_comm_get_unique_id(unsigned char *id);
int fv3_comm_init(fv3_ctx *ctx, int rank, int nranks, const unsigned char *id);
int fv3_comm_destroy(fv3_ctx *ctx);
int fv3_halo_start(fv3_ctx *ctx, int nfields, const fv3_halo_field *fields, const int *to, const int *from);
int fv3_halo_complete(fv3_ctx *ctx);
int fv3_allreduce_max(fv3_ctx *ctx, double *buf, int n);

/* The cube-edge exchange: mpp_update_domains on the six-tile mosaic of the cubed sphere (tools/fv_mp_mod.F90:498-546: the 12
 * contacts with index reversal and D / C-grid component rotation; group updates :646-876) and mpp_get_boundary of (u, v)
 * (model/dyn_core.F90:1151-1163), one face per rank (BASELINE config 5) or several faces per rank, as RCCL peer messages on the
 * communication stream of the first context: per pair of faces ONE message per call, whatever the number of fields.
 *   fv3_cube_table        the rows of the halo update of one member of one face: the library's own topology (csrc/cube_topo.h),
 *                         exposed so that a host can build gathers from it and tests can hold it to the oracle's contact-list
 *                         derivation.  Returns the row count (arrays may be NULL to count), -1 on a bad argument.
 *   fv3_cube_halo_start   pack (sign of the rotation applied unless scalar_pair), then the grouped sends / receives; kernels
 *                         launched on the contexts' streams before _complete overlap the transfers.  ctxs[0] carries the
 *                         communicator (fv3_comm_init); faces[] ascending; face_rank[6] = the rank holding each face;
 *                         fields[i * nfields + f] = field f of context i.  Everything is validated before the first pack.
 *   fv3_cube_halo_complete  the contexts' streams wait for the transfers, unpack into the halos. */
#define FV3_CUBE_A 0      /* cell centres                         f0: A x nk */
#define FV3_CUBE_B 1      /* corners                              f0: B x nk */
#define FV3_CUBE_D 2      /* D-grid pair                          f0 = u: U x nk, f1 = v: V x nk */
#define FV3_CUBE_C 3      /* C-grid pair                          f0 = uc: V x nk, f1 = vc: U x nk */
#define FV3_CUBE_DEDGE 4  /* mpp_get_boundary of the D-grid pair  f0 = u, f1 = v: u(:, npy), v(npx, :) from the neighbours */
typedef struct fv3_cube_field {
  int kind;
  double *f0, *f1;   /* device arrays, f1 NULL for kinds A / B */
  int nk;
  int scalar_pair;   /* 1: no sign change (flags = SCALAR_PAIR) */
} fv3_cube_field;
long fv3_cube_table(int npx, int ng, int kind, int member, int face, long *dst, int *src_face, int *comp, long *src, int *sign);
int fv3_cube_halo_start(int nctx, fv3_ctx *const *ctxs, const int *faces, const int *face_rank, int nfields,
                        const fv3_cube_field *fields);
int fv3_cube_halo_complete(int nctx, fv3_ctx *const *ctxs);

/* ---- nonhydrostatic column path --------------------------------------------------------------------
 * Physical constants live in FMS constants_mod (not part of the reference tree); the caller passes them. */
typedef struct fv3_nh_consts {
  double grav, rdgas, cp_air, akap, ptop, p_fac, a_imp;
  int m_split;   /* flagstruct%m_split: the sub-steps of RIM_2D (taken for a_imp <= 0.5, nh_core.F90:175, nh_utils.F90:452); >= 1 */
} fv3_nh_consts;

/* dp_ref(k) = ak(k+1)-ak(k) + (bk(k+1)-bk(k))*1e5 (model/dyn_core.F90:241-244), HOST array of length npz.
 * Also precomputes the level-only coefficients of edge_profile (model/nh_utils.F90:1640-1662). */
int fv3_set_dp_ref(fv3_ctx *ctx, const double *dp0);

/* update_dz_c -- model/nh_utils.F90:59, call site model/dyn_core.F90:525.  gz_in/gz: A x (npz+1) (the
 * reference advects gz in place after copying zh into it, dyn_core.F90:491-521; pass zh as gz_in);
 * ut, vt: A x npz (c_sw outputs); zs, ws: A.  Outputs valid on is-1:ie+1 x js-1:je+1. */
int fv3_update_dz_c(fv3_ctx *ctx, double dt, const double *zs, const double *ut, const double *vt,
                    const double *gz_in, double *gz, double *ws);

/* use_cond / moist_kappa of Riem_Solver_c (nh_utils.F90:383-396, :413-438) and Riem_Solver3 (nh_core.F90:96-166): the
 * device arrays q_con (A x npz, condensate mixing ratio: pm2 is then formed from the hydrostatic pressure without the
 * condensates) and cappa (A x npz, moist kappa: gm2 = 1/(1-cappa), cp2 = cappa per cell) that the following
 * fv3_riem_solver_c / fv3_riem_solver3 calls read.  NULL = .false. (the default).  As in the reference, Riem_Solver_c
 * uses cappa only together with q_con. */
int fv3_set_condensate(fv3_ctx *ctx, const double *q_con, const double *cappa);

/* (Rounds 2 - 4 had a second, tolerance mode of the column solvers -- fv3_set_fast / FV3_MI355X_FAST: blocked parallel scans, within 1e-12 per call
 * but 2e-12 in w after a whole dt_atmos.  Removed in round 5: one mode, the reference's elimination order, bit-comparable with the CPU oracle.) */

/* flagstruct%fast_tau_w_sec > 0: the Rayleigh damping of w inside SIM1_solver / SIM_solver (model/nh_utils.F90:1363-1371, :1498-1506;
 * the call sites hand the flag to both solvers, dyn_core.F90:536, :940).  rff: HOST array of k_rf values, the profile Riem_Solver_c
 * evaluates ONCE on its first call (nh_utils.F90:356-367: rff(k) = 1 / (1 + dt / fast_tau_w_sec * sin^2(...)) on the levels with
 * pfull <= rf_cutoff, with ITS dt -- half the acoustic step) and both solvers use from then on; the host evaluates it the same way
 * (gfdl_atmos_cubed_sphere_amd/dyn_core.py fast_tau_w_profile; oracle/nh_core.c fvo_fast_tau_w_rff) and hands it over once.  Every
 * fv3_riem_solver_c / fv3_riem_solver3 call of the context with a_imp > 0.5 then multiplies w2(k), k <= k_rf, by rff(k) behind the back
 * substitution.  k_rf = 0: off (the default). */
int fv3_set_fast_tau_w(fv3_ctx *ctx, int k_rf, const double *rff);

/* Ray_fast -- model/dyn_core.F90:2485-2601, call site :1057-1060 (flagstruct%RF_fast and tau > 0: at the end of every acoustic
 * substep).  fv3_set_ray_fast hands over what the routine keeps from its first call (:2519-2545): rf(1:kmax) = 1 / (1 + rff(k)) on the
 * levels with pfull < rf_cutoff, dp(1:npz) = dp_ref, k_rf and dm = sum of dp(1:k_rf) (HOST arrays, the host's arithmetic in the
 * reference's order).  fv3_ray_fast then scales u (U x npz), v (V x npz) and, unless hydrostatic, w (A x npz) on the levels k <= kmax by
 * rf(k) and gives the momentum a column lost, sum (1 - rf) dp u / dm, back to its levels k <= k_rf (:2549-2597), compute domain. */
int fv3_set_ray_fast(fv3_ctx *ctx, int kmax, int k_rf, double dm, const double *rf, const double *dp);
int fv3_ray_fast(fv3_ctx *ctx, double *u, double *v, double *w, int hydrostatic);
/* mix_dp -- model/dyn_core.F90:2119-2200, called behind the d_sw loop when flagstruct%fill_dp (:820, CG = .false.): on the compute domain a
 * layer with delp below 1 % of its reference thickness (0.01 (ak(k+1) - ak(k) + (bk(k+1) - bk(k)) 1e5); NaN counts) takes the missing mass
 * from the layer below (the bottom layer: from above) and mixes pt (and w unless hydrostatic) with it, top to bottom.  delp, pt, w: A x npz,
 * in place; needs fv3_set_ak_bk. */
int fv3_mix_dp(fv3_ctx *ctx, int hydrostatic, double *w, double *delp, double *pt);

/* Riem_Solver_c -- model/nh_utils.F90:323, call site model/dyn_core.F90:531 (a_imp > 0.5: SIM1_solver; a_imp < -0.01: SIM3p0_solver;
 * otherwise RIM_2D with cn->m_split sub-steps, nh_utils.F90:449-459).
 * hs, ws: A; w3 (=omga), pt (=ptc), delp (=delpc): A x npz; gz (in/out), pef (=pkc, out): A x (npz+1). */
int fv3_riem_solver_c(fv3_ctx *ctx, double dt, const fv3_nh_consts *cn, const double *hs, const double *w3,
                      const double *pt, const double *delp, double *gz, double *pef, const double *ws);

/* update_dz_d -- model/nh_utils.F90:204, call site model/dyn_core.F90:911.  Uses the per-level nord_v /
 * damp_vt uploaded with fv3_dsw_levels_upload (entry npz+1 = entry npz, nh_utils.F90:240-241).
 * zh_in -> zh_out (A x (npz+1), compute domain written; must not alias); crx, xfx: CX x npz; cry, yfx:
 * CY x npz; zs: A; ws: CC. */
int fv3_update_dz_d(fv3_ctx *ctx, int hord, const double *zs, const double *zh_in, double *zh_out,
                    const double *crx, const double *cry, const double *xfx, const double *yfx, double *ws,
                    double rdt);

/* Riem_Solver3 -- model/nh_core.F90:47, call site model/dyn_core.F90:932 (a_imp > 0.999: SIM1_solver, else
 * SIM_solver).  w, zh in/out; delz (CC x npz), ppe (=pkc), pk3 (A x (npz+1)) out; pe (is-1:ie+1, npz+1,
 * js-1:je+1), pk (CC x (npz+1)), peln (is:ie, npz+1, js:je) written when last_call. */
int fv3_riem_solver3(fv3_ctx *ctx, double dt, const fv3_nh_consts *cn, const double *zs, double *w, double *delz,
                     const double *pt, const double *delp, double *zh, double *pe, double *ppe, double *pk3,
                     double *pk, double *peln, const double *ws, int use_logp, int last_call, int fp_out);

/* p_grad_c -- model/dyn_core.F90:1635, call site :562.  uc (V x npz), vc (U x npz) updated in place. */
int fv3_p_grad_c(fv3_ctx *ctx, double dt2, const double *delpc, const double *pkc, const double *gz, double *uc,
                 double *vc, int hydrostatic);

/* nh_p_grad -- model/dyn_core.F90:1697, call site :1032.  u (U x npz), v (V x npz) updated in place (and
 * multiplied by rdx, rdy).  pp (=pkc), pk (=pk3), gz, delp are NOT modified (the reference overwrites them
 * with their corner interpolants, which nothing reads afterwards). top_value = ptk or peln1 (:1723-1727).
 * gz_scale: the kernel reads gz*gz_scale (pass zh and grav to fuse "gz = zh*grav", dyn_core.F90:982-989;
 * 1.0 for a plain gz). */
int fv3_nh_p_grad(fv3_ctx *ctx, double *u, double *v, const double *pp, const double *gz, double gz_scale,
                  const double *delp, const double *pk, double dt, double top_value);

/* split_p_grad -- model/dyn_core.F90:1795-1900, call site :1028 (beta > 0: the hydrostatic part of the pressure gradient split
 * between two substeps).  Arguments of fv3_nh_p_grad plus beta (the caller's beta_d: 0 in the first substep, :398-406) and du, dv
 * (U / V x npz, device, the caller's: zero before the first call as dyn_core.F90:278-283 allocates them; updated in place). */
int fv3_split_p_grad(fv3_ctx *ctx, double *u, double *v, const double *pp, const double *gz, double gz_scale, const double *delp,
                     const double *pk, double beta, double dt, double top_value, double *du, double *dv);

/* omega of the last acoustic substep, local part (model/dyn_core.F90:409-421, :1182-1191, use_old_omega):
 * omga(i,j,k) = (pe(i,k+1,j) - pem(i,k+1,j)) * rdt with pem = ptop + cumulative sum of the delp the substep started
 * from (pass the pre-d_sw buffer as delp_before).  The advective term adv_pe (:1195, :1529-1630) projects on the unit
 * vectors en1/en2, which the reference only sets for grid_type < 3 (fv_grid_utils.F90:628-643); for the grid_type = 4
 * domains built here that term is undefined in the reference and is not added. */
int fv3_omga_update(fv3_ctx *ctx, double rdt, double ptop, const double *pe, const double *delp_before, double *omga);

/* Hydrostatic pressure gradient.
 * divg2_ext -- model/dyn_core.F90:745-747, :791-797, :828-848: external-mode divergence damping field at the corners,
 *   divg2 = d_ext*da_min_c * sum_k ptc*vt / sum_k ptc with ptc = a2b_ord2(delp BEFORE d_sw) and vt = d_sw's delpc
 *   output; divg2: A-kind 2-D array addressed with corner indices (zeros when d_ext <= 0).
 * one_grad_p -- model/dyn_core.F90:1909, call site :1021, hydrostatic form (pk = pe**kappa, ptk = ptop**kappa):
 *   u, v updated in place (and multiplied by rdx, rdy); pk, gz (A x (npz+1)) are NOT modified (the reference
 *   replaces them by their corner interpolants, which nothing reads afterwards); divg2 may be NULL (d_ext <= 0).
 * copy_a_to_cc -- "pk = pkc" on the last substep (:1001-1010). */
/* adv_pe (model/dyn_core.F90:1195, :1529-1632): the advective term of omega on a cubed-sphere face, added to omga (A x npz) on
 * the compute domain after fv3_omga_update: om += 0.5*rarea * (V3 . grad pe) with pe at the corners by a2b_ord2 of pem, the edge
 * pressures of the state the last substep started from (delp_before with its halo, as in fv3_omga_update; pem is formed on
 * (is-1:ie+1, js-1:je+1)); ua, va: the A-grid winds c_sw left (A x npz).  Needs ec1 .. en2 of fv3_grid_cubed.  (On grid_type = 4
 * the reference leaves en1 / en2 unset: not defined there.) */
int fv3_adv_pe(fv3_ctx *ctx, double ptop, const double *ua, const double *va, const double *delp_before, double *omga);
int fv3_divg2_ext(fv3_ctx *ctx, double d_ext, const double *delp, const double *vt, double *divg2);
int fv3_one_grad_p(fv3_ctx *ctx, double *u, double *v, const double *pk, const double *gz, const double *divg2,
                   double dt, double ptk);
/* one_grad_p with hydrostatic = .false. -- model/dyn_core.F90:1909-2030, the call of the NONHYDROSTATIC loop with beta < -0.1 (:1029-1030)
 * after Riem_Solver3 left the full pressure in pkc (its `fp_out`, :939): pk(:,:,1) = ptop (:1950) and the layer weight is a2b_ord4 of
 * delp (:1996-1997) instead of the difference of the corner pk.  pk: the full pressure (A x (npz+1), halo filled), gz: the interface
 * heights times gz_scale (pass zh and grav: gz = zh * grav, :982-989), delp: A x npz with its halo, divg2: fv3_divg2_ext or null. */
int fv3_one_grad_p_nh(fv3_ctx *ctx, double *u, double *v, const double *pk, const double *gz, const double *divg2,
                      const double *delp, double dt, double ptop, double gz_scale);

/* grad1_p_update -- model/dyn_core.F90:2033-2116, call site :1019 (hydrostatic, beta > 0).  divg2: A (fv3_divg2_ext; zeros when
 * d_ext = 0), du, dv as for fv3_split_p_grad. */
int fv3_grad1_p_update(fv3_ctx *ctx, const double *divg2, double *u, double *v, const double *pk, const double *gz, double dt,
                       double ptk, double beta, double *du, double *dv);
int fv3_copy_a_to_cc(fv3_ctx *ctx, const double *src, double *dst, int nk);

/* zh(npz+1) = zs; zh(k) = zh(k+1) - delz(k) on the compute domain -- model/dyn_core.F90:370-385 (it == 1). */
int fv3_zh_from_delz(fv3_ctx *ctx, const double *zs, const double *delz, double *zh);

/* pk3_halo / pln_halo (use_logp) and pe_halo -- model/dyn_core.F90:1395,1449,1498; call sites :953-958. */
int fv3_pk3_halo(fv3_ctx *ctx, double ptop, double akap, double *pk3, const double *delp, int use_logp);
int fv3_pe_halo(fv3_ctx *ctx, double ptop, double *pe, const double *delp);

/* geopk -- model/dyn_core.F90:2202 (hydrostatic path; call sites :481 (CG=1), :906 (CG=0)).  ptk = ptop**akap. */
int fv3_geopk(fv3_ctx *ctx, double ptop, double akap, double cp_air, double ptk, double *pe, double *peln,
              const double *delp, double *pk, double *gz, const double *hs, const double *pt, double *pkz, int CG);

/* ---- dissipative heating after the substep loop (model/dyn_core.F90:798-803, :1300-1355) -------------------------
 * heat_source: A x npz (the caller zeroes it at the start of dyn_core, :294).
 * accum: heat_source += heat_s on the compute domain after every d_sw (d_con > 1e-5).
 * del2_cubed -- model/dyn_core.F90:2356: min(3, nmax) smoothing passes on shrinking boxes; the halo of q must be up
 *   to date on entry (the reference calls mpp_update_domains first, :2399); uses one context scratch slab.
 * apply: pt += sign(min(delt, |dT|), dT)/pkz etc. for k = 1..n_con; nonhydrostatic: pkz is recomputed from delp, delz,
 *   pt (:1347), with the exponent cappa/(1-cappa) when a cappa array is set (thermostruct%moist_kappa, :1338-1340:
 *   fv3_set_condensate -- the array the Riemann solvers use); delz, pkz: CC x npz. */
int fv3_heat_source_accum(fv3_ctx *ctx, double *heat_source, const double *heat_s);
int fv3_del2_cubed(fv3_ctx *ctx, double *q, int nk, double cd, int nmax);
int fv3_apply_heat_source(fv3_ctx *ctx, int n_con, int hydrostatic, double bdt, double delt_max, double cp_air,
                          double cv_air, double rdgas, double grav, double *pt, double *heat_source, const double *delp,
                          const double *delz, double *pkz);

/* ---- fv_dynamics around the k_split loop ---------------------------------------------------------------------------
 * fv3_c2l = cubed_to_latlon (model/fv_grid_utils.F90:2319): c2l_ord = 2 -> c2l_ord2 (:2526-2558), 4 -> c2l_ord4
 *   (:2384-2475; the halo update of u, v that the reference does first with mode > 0, :2372-2376, is the caller's); the
 *   Cartesian branches on grid_type = 4, on a cubed-sphere face the two-point forms next to the face edges and the rotation
 *   to (east, north) with a11 .. a22 of fv3_grid_cubed.
 *   u: U x npz, v: V x npz in; ua, va: A x npz out on the compute domain.  Called at fv_dynamics.F90:911.
 * Rayleigh_Friction (model/fv_dynamics.F90:1126-1264; the branch of :368-376 for grid_type = 4) is two calls around
 *   the halo update of u2f (:1207-1209), which the caller performs with its halo exchanger:
 *   fv3_rayleigh_u2f: ua, va by c2l_ord2 and u2f = ua^2 + va^2 (+ w^2) on levels 1..kmax (:1186-1205);
 *   fv3_rayleigh_apply: frictional heating of pt (and delz) if conserve, then u, v, w /= 1 + rf(k)*sqrt(u2f/4900)
 *   averaged to their points (:1211-1260).  pm, rf: HOST arrays of length kmax (layer-mean pressure and the damping
 *   profile of :1169-1182, which the caller evaluates once); u2f: A x kmax, read only; cp = cp_air, rg = rdgas. */
int fv3_c2l(fv3_ctx *ctx, int c2l_ord, const double *u, const double *v, double *ua, double *va);
/* flagstruct%consv_am -- model/fv_dynamics.F90:358-361 (before the k_split loop) and :747-800 (after it).
 * fv3_compute_aam = compute_aam (:1266-1314) behind the caller's fv3_c2l(ctx, 2, u, v, ua, va) (:1287): per column of the compute domain
 *   aam = sum (r^2 omega + r ua) dm, m_fac = sum dm r^2 (dm = delp * agrav, r = radius * coslat) and ps = ptop + sum delp.
 *   coslat: A (2-D) = cos(agrid(:,:,2)) (the caller's, as gridstruct%agrid is); ua, delp: A x npz; aam, m_fac: CC; ps: A (2-D).
 *   The caller forms te_2d - teq + dt2 (ps2 + ps) zxg, the two reproducible g_sums and u00 (:761-776) on these small 2-D fields.
 * fv3_consv_am_apply (:784-798): u += u00 l2c_u on (is:ie, js:je+1), v += u00 l2c_v on (is:ie+1, js:je), every level;
 *   l2c_u: U (2-D), l2c_v: V (2-D) = gridstruct%l2c_u / l2c_v (fv_grid_utils.F90:402-424) on the device. */
int fv3_compute_aam(fv3_ctx *ctx, double radius, double omega, double agrav, double ptop, const double *coslat, const double *ua,
                    const double *delp, double *aam, double *m_fac, double *ps);
int fv3_consv_am_apply(fv3_ctx *ctx, double u00, const double *l2c_u, const double *l2c_v, double *u, double *v);
int fv3_rayleigh_u2f(fv3_ctx *ctx, int kmax, int hydrostatic, const double *u, const double *v, const double *w,
                     double *ua, double *va, double *u2f);
int fv3_rayleigh_apply(fv3_ctx *ctx, int kmax, int conserve, int hydrostatic, double cp, double rg, double ptop,
                       const double *pm, const double *rf, const double *u2f, double *pt, double *delz, double *u,
                       double *v, double *w);
/* Rayleigh_Super (model/fv_dynamics.F90:953-1124; the branch of :368-376 for grid_type < 4, bounded domains and ideal
 * cases) after the caller's fv3_c2l(ctx, 2, u, v, ua, va) (:1040-1042): on levels 1..kmax the damping factor is the
 * level constant 1 / (1 + rf(k)) (its halo update, :1054, moves a constant), so this is ONE call: frictional heating of
 * pt from ua, va (, w) if conserve (:1084-1098), then u, v, w scaled (:1100-1117); with u00, v00 (is_ideal_case, the
 * t = 0 winds the routine keeps, U / V x npz) the relaxation towards them instead (:1064-1081).  pm, rf: HOST arrays of
 * length kmax as for fv3_rayleigh_apply; w may be NULL when hydrostatic; u00 / v00 NULL = not an ideal case. */
int fv3_rayleigh_super(fv3_ctx *ctx, int kmax, int conserve, int hydrostatic, double cp, double rg, double ptop,
                       const double *pm, const double *rf, const double *ua, const double *va, double *pt, double *u,
                       double *v, double *w, const double *u00, const double *v00);

/* ---- fv_dynamics: T -> theta_v before the k_split loop (model/fv_dynamics.F90:284-329, :379-399; use_cond =
 * moist_kappa = .false. unless fv3_set_moist says otherwise).  nonhydrostatic: pkz = exp(kappa*log(rdg*delp*pt*(1+zvir*qv)/delz)) is (re)computed;
 * hydrostatic: pkz is taken as given (p_var / the previous remap).  Then pt = pt*(1+zvir*qv)/pkz on the compute
 * domain.  hydrostatic = -1: only pkz is computed and pt is left alone -- the reference evaluates pkz (:323-326) before
 * Rayleigh_Friction changes T and delz and converts afterwards with that pkz (:389-397): call with -1, apply the
 * friction, call with 1.  qv: A x npz specific humidity or NULL (dry: zvir*qv = 0).  The way back (theta_v -> T) is part of
 * fv3_lagrangian_to_eulerian with last_step = 1, as in the reference (fv_mapz.F90:793-821). */
int fv3_pt_to_theta_v(fv3_ctx *ctx, int hydrostatic, double zvir, double kappa, double rdgas, double grav, double *pt,
                      const double *delp, const double *delz, const double *qv, double *pkz);

/* ---- table-driven halo gather -------------------------------------------------------------------------------
 * The face-to-face halo updates of the cubed sphere (mpp_update_domains on the six-tile mosaic, contacts
 * tools/fv_mp_mod.F90:498-546: index reversal along an edge, u <-> v with sign for DGRID_NE / CGRID_NE pairs) as one
 * table per field kind: entry e copies  ptrs[dst_sel[e]][k*stride + dst_idx[e]] = sign[e] * ptrs[src_sel[e]][k*stride +
 * src_idx[e]]  for k = 0..nk-1.  Up to 16 array pointers per run (six faces x two members of a vector pair, or message
 * buffers: the same tables drive pack and unpack across GPUs).  Tables are HOST int arrays, copied once. */
typedef struct fv3_gather fv3_gather;
int fv3_gather_create(fv3_ctx *ctx, int n, const int *dst_sel, const int *dst_idx, const int *src_sel, const int *src_idx,
                      const int *sign, fv3_gather **out);
int fv3_gather_run(fv3_ctx *ctx, const fv3_gather *t, int nk, int nptr, double *const *ptrs, const size_t *strides);
int fv3_gather_destroy(fv3_gather *t);

/* ---- vertical remap ------------------------------------------------------------------------------------
 * Lagrangian_to_Eulerian -- model/fv_mapz.F90:56-64, call site model/fv_dynamics.F90:607.  Branches built:
 * remap_te=.false., consv=0, kord <= 15 (8..15: scalar_profile / cs_profile; <= 7, the signed value as the map routines test
 * it: ppm_profile + ppm_limiters, fv_operators.F90:1382-1723), kord_wz>0; use_cond / moist_kappa through fv3_set_moist below,
 * flagstruct%fill through fv3_remap_params.fill.
 * All fields are updated in place (every column is independent): ps (A), pe (is-1:ie+1, npz+1, js-1:je+1),
 * delp, pt, w, omga (A x npz), q (A x npz x nq), u (U x npz), v (V x npz), delz, pkz (CC x npz),
 * pk (CC x (npz+1)), peln (is:ie, npz+1, js:je); ws (CC, in).  On return pt is theta_v again
 * (or T_v when last_step), exactly as the reference leaves it. */
typedef struct fv3_remap_params {
  int last_step, hydrostatic, adiabatic, nq, kord_mt, kord_wz, kord_tm, sphum;
  double akap, ptop, rdgas, grav, cv_air, r_vir, cp, t_min;
  int fill;   /* flagstruct%fill: fillz (fv_fill.F90:34-137) on every remapped tracer column (fv_operators.F90:337) */
} fv3_remap_params;
/* thermostruct%moist_kappa / use_cond in the remap (nonhydrostatic): the T_v <-> T_m transforms use the moist kappa
 * cappa = rdgas / (rdgas + cvm/(1 + r_vir*qv)) with cvm, q_con from moist_cv (fv_thermodynamics.F90:250-325; nwat and the
 * 1-based tracer indices of the water species, 0 = absent; cv_vap = 3*rvgas, c_liq, c_ice of gfdl_mp.F90:136-137), q_con
 * and cappa (A x npz, device) are rewritten (fv_mapz.F90:212-219, :463-478), and with use_cond the last step returns
 * T = T_m / ((1 + r_vir*qv)(1 - q_con)) (:806-811).  fv3_pt_to_theta_v follows the same switches (fv_dynamics.F90:305-317,
 * :381-388): with moist_kappa its qv argument must be &q(isd,jsd,1,sphum) of the full tracer array, pkz uses cappa and
 * (1 - q_con) and both arrays are written; with use_cond the conversion carries (1 - q_con).  m = NULL switches the moist
 * branches off (the default). */
typedef struct fv3_moist_params {
  int moist_kappa, use_cond, nwat, sphum, liq_wat, rainwat, ice_wat, snowwat, graupel;
  double cv_vap, c_liq, c_ice;
} fv3_moist_params;
int fv3_set_moist(fv3_ctx *ctx, const fv3_moist_params *m, double *q_con, double *cappa);
/* flagstruct%remap_te (model/fv_arrays.F90:399; fv_mapz.F90:232-286, :348-360, :576-619, :655-663): the remap carries the total
 * energy cp T + KE + phis of every layer (map_scalar with abs(kord_tm) in log p, or map1_cubic when kord_tm = 0) in the place of
 * T_v / theta_v, and T_v and pkz follow from it after the winds were remapped.  hs = phis (A), te: A x npz work array, both device
 * (the reference's hs and te arguments; fv_dynamics hands it dp1 as te).  fv3_energy_fixer_sums then takes te_2d from te.  The
 * reference's j loop reads the winds of row j + 1 before it remaps them; the library keeps a copy of u to do the same. */
int fv3_set_remap_te(fv3_ctx *ctx, int remap_te, const double *hs, double *te);
/* ak, bk: HOST arrays of length npz+1 (the hybrid coordinate, tools/fv_eta.F90). */
int fv3_set_ak_bk(fv3_ctx *ctx, const double *ak, const double *bk);
/* ---- total-energy conservation (consv_te) -------------------------------------------------------------------------
 * fv3_compute_total_energy = compute_total_energy (model/fv_thermodynamics.F90:90-225; called at fv_dynamics.F90:345 before
 *   the T -> theta_v conversion): te_2d (CC) of every column; qc = zvir*q(sphum) (A x npz, or NULL = 0), q the tracers
 *   (read by the moist_phys .and. moist_kappa branch only), w / delz or pe / peln by branch.  The scalars come from
 *   the fv3_remap_params structure: hydrostatic, rdgas, cp, cv_air, grav, nq, sphum, the moist switches from fv3_set_moist.
 * The energy fixer of the last remap (model/fv_mapz.F90:643-772, :793-821) is three steps around the caller's global sums:
 *   fv3_lagrangian_to_eulerian with last_step = 2 -- the remap of the last step WITHOUT its final T_v -> T conversion;
 *   fv3_energy_fixer_sums -- te_2d := te0_2d - (total energy of the remapped columns) (:647-722), zsum1 = sum_k pkz*delp,
 *     zsum0 = ptop*(pk(1) - pk(npz+1)) + zsum1 (hydrostatic) (:723-734), all CC; only_sums = 1: the sums alone (the
 *     consv < -consv_min branch, :745-763); with use_cond q_con is rewritten from moist_cv (:701);
 *   [caller: dtmp = consv * g_sum(te_2d) / g_sum(zsum0 | zsum1), g_sum = area-weighted reproducing sum, :736-743]
 *   fv3_remap_finish(dtmp) -- pt = (pt + dtmp/c * pkz) / (1 + r_vir*qv) in the three forms of :793-821 (c = cp, cvm, cv_air;
 *     nonhydrostatic adiabatic: nothing).  With adiabatic set the virtual factor is 1 (the reference's caller passes zvir = 0).
 *   remap_te = .true. is not built. */
/* g_sum(domain, p, ..., area, mode = 0, reproduce = .true.) (model/fv_grid_utils.F90:2879-2925; = FMS mpp_global_sum with
 * BITWISE_EFP_SUM): the reproducing sum of n HOST doubles (the caller's p*area of its compute domain) over the ranks of the
 * context's communicator (ctx may be NULL / without a communicator: one rank).  Extended fixed point: integer digits, exact,
 * independent of the order of the addends and of the rank layout. */
int fv3_ordered_sum(fv3_ctx *ctx, const double *values, size_t n, double *sum);
int fv3_compute_total_energy(fv3_ctx *ctx, const fv3_remap_params *p, int moist_phys, const double *u, const double *v,
                             const double *w, const double *delz, const double *pt, const double *delp, const double *q,
                             const double *qc, const double *pe, const double *peln, const double *phis, double *te_2d);
int fv3_energy_fixer_sums(fv3_ctx *ctx, const fv3_remap_params *p, int only_sums, const double *u, const double *v,
                          const double *w, const double *delz, const double *pt, const double *delp, const double *q,
                          const double *pe, const double *peln, const double *phis, const double *pkz, const double *pk,
                          const double *te0_2d, double *te_2d, double *zsum1, double *zsum0);
int fv3_remap_finish(fv3_ctx *ctx, const fv3_remap_params *p, double dtmp, double *pt, const double *pkz, const double *q);

int fv3_lagrangian_to_eulerian(fv3_ctx *ctx, const fv3_remap_params *p, const int *kord_tr, double *ps, double *pe,
                               double *delp, double *pkz, double *pk, double *u, double *v, double *w, double *delz,
                               double *pt, double *q, double *peln, double *omga, const double *ws);

/* ---- tracer_2d -- model/fv_tracer2d.F90:297, call site model/fv_dynamics.F90:531 -------------------------
 * The routine is split at the point where it needs a cross-rank reduction (mp_reduce_max, :405): the caller
 * (host) runs prep, reduces cmax(npz) over ranks, chooses nsplt / ksplt(k) / frac(k) exactly as :404-430 do,
 * then scale (if nsplt != 1) and nsplt times [halo update of q; step].
 * q, q_out: A x npz x nq; dp1, dp1_out: A x npz; mfx: FX x npz; mfy: FY x npz; cx, xfx: CX x npz; cy, yfx: CY x npz. */
int fv3_tracer_2d_prep(fv3_ctx *ctx, int q_split, const double *cx, const double *cy, double *xfx, double *yfx,
                       double *cmax_host /* npz, out */);
int fv3_tracer_2d_scale(fv3_ctx *ctx, const double *frac_host /* npz */, double *cx, double *xfx, double *mfx,
                        double *cy, double *yfx, double *mfy);
/* one sub-cycle `it` (1-based) of nsplt; q -> q_out and (if it != nsplt) dp1 -> dp1_out on the compute domain.
 * hord (= hord_tr): as fv3_fv_tp_2d. */
int fv3_tracer_2d_step(fv3_ctx *ctx, int it, int nsplt, const int *ksplt_host /* npz */, int nq, int hord,
                       int nord_tr, double trdm, const double *q, double *q_out, const double *dp1, double *dp1_out,
                       const double *mfx, const double *mfy, const double *cx, const double *cy, const double *xfx,
                       const double *yfx);

/* inline_q -- model/sw_core.F90:1020-1043 (flagstruct%inline_q, dyn_core.F90:340, :573, :768): the tracers advected inside d_sw
 * every acoustic substep instead of tracer_2d on the accumulated fluxes.  Called after fv3_d_sw of the same substep with what that
 * call produced: crx, cry, xfx, yfx (crx_adv .. yfx_adv), fx / fy = the delp fluxes OF THIS SUBSTEP (hand fv3_d_sw zeroed arrays
 * as its mfx / mfy and add them to the accumulators with fv3_flux_accum), delp_old / delp_new = d_sw's delp and delp_out.  q (halo
 * updated, dyn_core.F90:341 / :573) -> q_out on the compute domain, A x npz x nq.  nord_t, damp_t: dyn_core.F90:690-692
 * (min(2, nord), vtdm4 with do_vort_damp): deln_flux with mass = d_sw's half-updated delp, sw_core.F90:1034. */
int fv3_d_sw_inline_q(fv3_ctx *ctx, int nq, int hord_tr, int nord_t, double damp_t, const double *q, double *q_out,
                      const double *delp_old, const double *delp_new, const double *fx, const double *fy, const double *crx,
                      const double *cry, const double *xfx, const double *yfx);
/* mfx += fx, mfy += fy (FX / FY x npz): sw_core.F90:949-962 for a d_sw that wrote its fluxes into arrays of their own */
int fv3_flux_accum(fv3_ctx *ctx, double *mfx, double *mfy, const double *fx, const double *fy);

/* fill2D -- model/fv_fill.F90:183-258, call site fv_dynamics.F90:542-556 (FILL2D builds; hord_tr < 8 and moist_phys): the
 * diffusive filling of negative tracer mass, one tracer (A x nk) per call pair.  _mass: qt = q * delp * area on the compute
 * domain (:228-235); the caller updates the halo of qt (width 1, :236); _apply: the sign-change fluxes and the update of q
 * (:238-256). */
int fv3_fill2d_mass(fv3_ctx *ctx, int nk, const double *q, const double *delp, double *qt);
int fv3_fill2d_apply(fv3_ctx *ctx, int nk, const double *qt, const double *delp, double *q);

/* Per-kernel timing with HIP events recorded on the context's stream around every kernel the
 * library launches (this is what bench.py's roofline figures are measured with).  report: one line
 * "label count total_ms" per kernel label since the last report; synchronises the stream. */
int fv3_profile(fv3_ctx *ctx, int enable);
int fv3_profile_report(fv3_ctx *ctx, char *out, size_t cap);
/* The same events under the reference's timer names (timing_on / timing_off: C_SW, D_SW, UPDATE_DZ_C, UPDATE_DZ, Riem_Solver, PG_D,
 * COMM_TOTAL, tracer_2d, Fill2D, Remapping -- model/dyn_core.F90:437-1015, fv_dynamics.F90:521-571 -- and DYN_CORE for what
 * dyn_core runs outside its inner timers): "timer count total_ms" per line.  Either report consumes the events. */
int fv3_profile_report_timers(fv3_ctx *ctx, char *out, size_t cap);

/* prt_maxmin / prt_mxm -- tools/fv_diagnostics.F90:4213-4313: out = {max * fac, min * fac, gmean * fac} of an A-kind field of nk
 * levels over the compute domain; gmean is prt_mxm's g_sum(q(:,:,nk), area, mode = 1) (fv_grid_utils.F90:2879-2925).  Reduced
 * over the ranks of the context's communicator (mp_reduce_min / _max / _sum).  Synchronises the stream. */
int fv3_prt_maxmin(fv3_ctx *ctx, const double *q, int nk, double fac, double out[3]);

#ifdef __cplusplus
}
#endif
#endif
