!> ISO_C_BINDING interface to libfv3_mi355x.so (include/fv3_mi355x.h) -- the binding a maintainer of
!> the reference would add so that model/dyn_core.F90 can call the MI355X kernels where it calls
!> c_sw / d_sw / fv_tp_2d today.  Device buffers are held as type(c_ptr); scalars go by value.
!> Compile check: amdflang -c fv3_mi355x_mod.F90   (no FMS needed).
module fv3_mi355x_mod
  use iso_c_binding
  implicit none
  private
  public :: fv3_domain, fv3_grid_host, fv3_dsw_params, fv3_dsw_levels
  public :: fv3_create, fv3_destroy, fv3_set_stream, fv3_grid_upload, fv3_malloc, fv3_free
  public :: fv3_group_create, fv3_group_flush, fv3_group_stats, fv3_group_destroy
  public :: fv3_memcpy_h2d, fv3_memcpy_d2h, fv3_sync, fv3_c_sw, fv3_d_sw, fv3_fv_tp_2d, fv3_ppm_line
  public :: fv3_dsw_levels_upload, fv3_halo_fill_periodic, fv3_check
  public :: fv3_nh_consts, fv3_remap_params, fv3_memcpy_d2d, fv3_set_dp_ref, fv3_update_dz_c, fv3_riem_solver_c
  public :: fv3_update_dz_d, fv3_riem_solver3, fv3_p_grad_c, fv3_nh_p_grad, fv3_zh_from_delz, fv3_pk3_halo
  public :: fv3_pe_halo, fv3_geopk, fv3_set_ak_bk, fv3_lagrangian_to_eulerian, fv3_tracer_2d_prep
  public :: fv3_tracer_2d_scale, fv3_tracer_2d_step, fv3_grid_geom
  public :: fv3_halo_field, fv3_halo_message_elems, fv3_halo_pack, fv3_halo_unpack, fv3_halo_periodic_group
  public :: fv3_heat_source_accum, fv3_del2_cubed, fv3_apply_heat_source
  public :: fv3_d_sw_interior, fv3_d_sw_rest
  public :: fv3_divg2_ext, fv3_one_grad_p, fv3_one_grad_p_nh, fv3_grad1_p_update, fv3_split_p_grad, fv3_d_sw_inline_q, fv3_set_remap_te, fv3_profile_report_timers, fv3_prt_maxmin, fv3_flux_accum, fv3_fill2d_mass, fv3_fill2d_apply, fv3_copy_a_to_cc, fv3_pt_to_theta_v, fv3_omga_update
  public :: fv3_grid_cubed, fv3_grid_upload_cubed, fv3_gather_create, fv3_gather_run, fv3_gather_destroy
  public :: fv3_comm_get_unique_id, fv3_comm_init, fv3_comm_destroy, fv3_halo_start, fv3_halo_complete, fv3_allreduce_max
  public :: fv3_cube_field, fv3_cube_table, fv3_cube_halo_start, fv3_cube_halo_complete
  public :: FV3_CUBE_A, FV3_CUBE_B, FV3_CUBE_D, FV3_CUBE_C, FV3_CUBE_DEDGE
  public :: fv3_c2l, fv3_rayleigh_u2f, fv3_rayleigh_apply, fv3_rayleigh_super, fv3_compute_total_energy, fv3_energy_fixer_sums, fv3_remap_finish, fv3_ordered_sum, fv3_adv_pe, fv3_set_condensate, fv3_registry_mode, fv3_registry_put, fv3_registry_get, fv3_registry_host_touched, fv3_registry_fetch, fv3_registry_forget, fv3_registry_stats, fv3_set_fast_tau_w, fv3_set_ray_fast, fv3_ray_fast, fv3_mix_dp, fv3_compute_aam, fv3_consv_am_apply, fv3_set_moist, fv3_moist_params

  type, bind(C) :: fv3_domain
    integer(c_int) :: is, ie, js, je, ng, npx, npy, npz, grid_type
    integer(c_int) :: do_diss_est, prevent_diss_cooling, stretched_grid
    real(c_double) :: lim_fac
  end type

  type, bind(C) :: fv3_grid_host      ! host addresses (c_loc) of the gridstruct members
    real(c_double) :: da_min, da_min_c
    type(c_ptr) :: area, rarea, dxa, dya, rdxa, rdya, cosa_s, rsin2, f0
    type(c_ptr) :: dx, rdx, dyc, rdyc, cosa_v, sina_v, rsin_v, divg_u, del6_u
    type(c_ptr) :: dy, rdy, dxc, rdxc, cosa_u, sina_u, rsin_u, divg_v, del6_v
    type(c_ptr) :: rarea_c, fC, cosa, sina
    type(c_ptr) :: sin_sg, cos_sg
  end type

  type, bind(C) :: fv3_grid_cubed     ! extra members of a cubed-sphere face (grid_type < 3): host addresses + factors
    type(c_ptr) :: edge_w, edge_e, edge_s, edge_n, rsina
    real(c_double) :: corner_f(12)
    type(c_ptr) :: a11 = c_null_ptr, a12 = c_null_ptr, a21 = c_null_ptr, a22 = c_null_ptr   ! cubed_to_latlon matrix (A layout)
    type(c_ptr) :: ec1 = c_null_ptr, ec2 = c_null_ptr, en1 = c_null_ptr, en2 = c_null_ptr   ! adv_pe's unit vectors (3 planes each)
  end type

  !> one field (or vector pair) of a cube-edge exchange group (fv3_cube_halo_start)
  integer(c_int), parameter :: FV3_CUBE_A = 0, FV3_CUBE_B = 1, FV3_CUBE_D = 2, FV3_CUBE_C = 3, FV3_CUBE_DEDGE = 4
  type, bind(C) :: fv3_cube_field
    integer(c_int) :: kind
    type(c_ptr) :: f0, f1 = c_null_ptr
    integer(c_int) :: nk
    integer(c_int) :: scalar_pair = 0
  end type

  type, bind(C) :: fv3_dsw_params
    real(c_double) :: dt
    integer(c_int) :: hord_tr, hord_mt, hord_vt, hord_tm, hord_dp
    real(c_double) :: dddmp, d4_bg, kgb
    integer(c_int) :: hydrostatic, use_cond
  end type

  type, bind(C) :: fv3_dsw_levels     ! host arrays of length npz (dyn_core.F90:666-733)
    type(c_ptr) :: nord_k, nord_v, nord_w, nord_t
    type(c_ptr) :: d2_divg, damp_vt, damp_w, damp_t, d_con_k
  end type

  type, bind(C) :: fv3_halo_field     ! one field of a halo-update group (kind 0=A, 1=U, 2=V, 3=B)
    type(c_ptr) :: field
    integer(c_int) :: kind, nk
  end type

  type, bind(C) :: fv3_nh_consts      ! FMS constants_mod values + namelist scalars
    real(c_double) :: grav, rdgas, cp_air, akap, ptop, p_fac, a_imp
    integer(c_int) :: m_split = 1       ! flagstruct%m_split: the sub-steps of RIM_2D (a_imp <= 0.5)
  end type

  type, bind(C) :: fv3_moist_params   ! moist_kappa / use_cond of the remap + the inputs of moist_cv
    integer(c_int) :: moist_kappa, use_cond, nwat, sphum, liq_wat, rainwat, ice_wat, snowwat, graupel
    real(c_double) :: cv_vap, c_liq, c_ice
  end type

  type, bind(C) :: fv3_remap_params   ! Lagrangian_to_Eulerian scalars (fv_mapz.F90:56-64)
    integer(c_int) :: last_step, hydrostatic, adiabatic, nq, kord_mt, kord_wz, kord_tm, sphum
    real(c_double) :: akap, ptop, rdgas, grav, cv_air, r_vir, cp, t_min
    integer(c_int) :: fill          ! flagstruct%fill: fillz on the remapped tracers
  end type

  interface
    integer(c_int) function fv3_create(dom, ctx) bind(C, name="fv3_create")
      import :: c_int, c_ptr, fv3_domain
      type(fv3_domain), intent(in) :: dom
      type(c_ptr), intent(out) :: ctx
    end function
    integer(c_int) function fv3_destroy(ctx) bind(C, name="fv3_destroy")
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
    end function
    integer(c_int) function fv3_set_stream(ctx, stream) bind(C, name="fv3_set_stream")
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx, stream
    end function
    !> the faces (tiles) this rank holds, launched together: one kernel launch for all of them (include/fv3_mi355x.h)
    integer(c_int) function fv3_group_create(members, n, grp) bind(C, name="fv3_group_create")
      import :: c_int, c_ptr
      type(c_ptr), intent(in) :: members(*)
      integer(c_int), value :: n
      type(c_ptr), intent(out) :: grp
    end function
    integer(c_int) function fv3_group_flush(grp) bind(C, name="fv3_group_flush")
      import :: c_int, c_ptr
      type(c_ptr), value :: grp
    end function
    integer(c_int) function fv3_group_stats(grp, merged, single) bind(C, name="fv3_group_stats")
      import :: c_int, c_ptr, c_long
      type(c_ptr), value :: grp
      integer(c_long), intent(out) :: merged, single
    end function
    integer(c_int) function fv3_group_destroy(grp) bind(C, name="fv3_group_destroy")
      import :: c_int, c_ptr
      type(c_ptr), value :: grp
    end function
    integer(c_int) function fv3_grid_upload(ctx, g) bind(C, name="fv3_grid_upload")
      import :: c_int, c_ptr, fv3_grid_host
      type(c_ptr), value :: ctx
      type(fv3_grid_host), intent(in) :: g
    end function
    integer(c_int) function fv3_grid_upload_cubed(ctx, g) bind(C, name="fv3_grid_upload_cubed")
      import :: c_int, c_ptr, fv3_grid_cubed
      type(c_ptr), value :: ctx
      type(fv3_grid_cubed), intent(in) :: g
    end function
    ! table-driven halo gather (the cubed-sphere face-to-face updates): host int tables, device array pointers
    integer(c_int) function fv3_gather_create(ctx, n, dst_sel, dst_idx, src_sel, src_idx, sgn, handle) &
        bind(C, name="fv3_gather_create")
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
      integer(c_int), value :: n
      integer(c_int), intent(in) :: dst_sel(*), dst_idx(*), src_sel(*), src_idx(*), sgn(*)
      type(c_ptr), intent(out) :: handle
    end function
    integer(c_int) function fv3_gather_run(ctx, handle, nk, nptr, ptrs, strides) bind(C, name="fv3_gather_run")
      import :: c_int, c_ptr, c_size_t
      type(c_ptr), value :: ctx, handle
      integer(c_int), value :: nk, nptr
      type(c_ptr), intent(in) :: ptrs(*)
      integer(c_size_t), intent(in) :: strides(*)
    end function
    integer(c_int) function fv3_gather_destroy(handle) bind(C, name="fv3_gather_destroy")
      import :: c_int, c_ptr
      type(c_ptr), value :: handle
    end function
    integer(c_int) function fv3_grid_geom(ctx) bind(C, name="fv3_grid_geom")
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
    end function
    integer(c_int) function fv3_malloc(dptr, bytes) bind(C, name="fv3_malloc")
      import :: c_int, c_ptr, c_size_t
      type(c_ptr), intent(out) :: dptr
      integer(c_size_t), value :: bytes
    end function
    integer(c_int) function fv3_free(dptr) bind(C, name="fv3_free")
      import :: c_int, c_ptr
      type(c_ptr), value :: dptr
    end function
    integer(c_int) function fv3_memcpy_h2d(ctx, dst, src, bytes) bind(C, name="fv3_memcpy_h2d")
      import :: c_int, c_ptr, c_size_t
      type(c_ptr), value :: ctx, dst, src
      integer(c_size_t), value :: bytes
    end function
    integer(c_int) function fv3_memcpy_d2h(ctx, dst, src, bytes) bind(C, name="fv3_memcpy_d2h")
      import :: c_int, c_ptr, c_size_t
      type(c_ptr), value :: ctx, dst, src
      integer(c_size_t), value :: bytes
    end function
    integer(c_int) function fv3_sync(ctx) bind(C, name="fv3_sync")
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
    end function
    !> replaces: call c_sw(delpc, delp, ptc, pt, u, v, w, uc, vc, ua, va, wc, ut, vt, divg_d, nord, dt2,
    !>                     hydrostatic, dord4, bd, gridstruct, flagstruct)   sw_core.F90:79, dyn_core.F90:439-447
    integer(c_int) function fv3_c_sw(ctx, delpc, delp, ptc, pt, u, v, w, uc, vc, ua, va, wc, ut, vt, divg_d, &
                                     nord, dt2, hydrostatic, dord4) bind(C, name="fv3_c_sw")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx, delpc, delp, ptc, pt, u, v, w, uc, vc, ua, va, wc, ut, vt, divg_d
      integer(c_int), value :: nord, hydrostatic, dord4
      real(c_double), value :: dt2
    end function
    integer(c_int) function fv3_dsw_levels_upload(ctx, lv) bind(C, name="fv3_dsw_levels_upload")
      import :: c_int, c_ptr, fv3_dsw_levels
      type(c_ptr), value :: ctx
      type(fv3_dsw_levels), intent(in) :: lv
    end function
    !> replaces: call d_sw(...)   sw_core.F90:494, dyn_core.F90:762-772
    integer(c_int) function fv3_d_sw(ctx, p, delpc, delp, pt, u, v, w, uc, vc, ua, va, divg_d, mfx, mfy, cx, cy, &
                                     crx, cry, xfx, yfx, q_con, delp_out, pt_out, u_out, v_out, w_out, q_con_out, &
                                     heat_s, diss_e) bind(C, name="fv3_d_sw")
      import :: c_int, c_ptr, fv3_dsw_params
      type(c_ptr), value :: ctx
      type(fv3_dsw_params), intent(in) :: p
      type(c_ptr), value :: delpc, delp, pt, u, v, w, uc, vc, ua, va, divg_d, mfx, mfy, cx, cy, crx, cry, xfx, yfx
      type(c_ptr), value :: q_con, delp_out, pt_out, u_out, v_out, w_out, q_con_out, heat_s, diss_e
    end function
    integer(c_int) function fv3_d_sw_interior(ctx, p, delpc, delp, pt, u, v, w, uc, vc, ua, va, divg_d, mfx, mfy, cx, cy, &
                                     crx, cry, xfx, yfx, q_con, delp_out, pt_out, u_out, v_out, w_out, q_con_out, &
                                     heat_s, diss_e) bind(C, name="fv3_d_sw_interior")
      import :: c_int, c_ptr, fv3_dsw_params
      type(c_ptr), value :: ctx
      type(fv3_dsw_params), intent(in) :: p
      type(c_ptr), value :: delpc, delp, pt, u, v, w, uc, vc, ua, va, divg_d, mfx, mfy, cx, cy, crx, cry, xfx, yfx
      type(c_ptr), value :: q_con, delp_out, pt_out, u_out, v_out, w_out, q_con_out, heat_s, diss_e
    end function
    integer(c_int) function fv3_d_sw_rest(ctx, p, delpc, delp, pt, u, v, w, uc, vc, ua, va, divg_d, mfx, mfy, cx, cy, &
                                     crx, cry, xfx, yfx, q_con, delp_out, pt_out, u_out, v_out, w_out, q_con_out, &
                                     heat_s, diss_e) bind(C, name="fv3_d_sw_rest")
      import :: c_int, c_ptr, fv3_dsw_params
      type(c_ptr), value :: ctx
      type(fv3_dsw_params), intent(in) :: p
      type(c_ptr), value :: delpc, delp, pt, u, v, w, uc, vc, ua, va, divg_d, mfx, mfy, cx, cy, crx, cry, xfx, yfx
      type(c_ptr), value :: q_con, delp_out, pt_out, u_out, v_out, w_out, q_con_out, heat_s, diss_e
    end function
    !> replaces: call fv_tp_2d(q, crx, cry, npx, npy, hord, fx, fy, xfx, yfx, gridstruct, bd, ra_x, ra_y, lim_fac,
    !>                         mfx, mfy, mass, nord, damp_c)   tp_core.F90:85
    integer(c_int) function fv3_fv_tp_2d(ctx, nk, q, crx, cry, hord, fx, fy, xfx, yfx, ra_x, ra_y, mfx, mfy, mass, &
                                         nord, damp_c) bind(C, name="fv3_fv_tp_2d")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx, q, crx, cry, fx, fy, xfx, yfx, ra_x, ra_y, mfx, mfy, mass
      integer(c_int), value :: nk, hord, nord
      real(c_double), value :: damp_c
    end function
    !> xppm / yppm (tp_core.F90:324-1152, private there) on one line with 3 halo cells either side: a unit-test surface
    integer(c_int) function fv3_ppm_line(ctx, iord, which, h, c, flux, n) bind(C, name="fv3_ppm_line")
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx, h, c, flux
      integer(c_int), value :: iord, which, n
    end function
    !> replaces start/complete_group_halo_update on a single-rank doubly periodic tile (fv_mp_mod.F90:646-876)
    integer(c_int) function fv3_halo_fill_periodic(ctx, field, kind, nk) bind(C, name="fv3_halo_fill_periodic")
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx, field
      integer(c_int), value :: kind, nk
    end function
    function fv3_last_error() bind(C, name="fv3_last_error") result(msg)
      import :: c_ptr
      type(c_ptr) :: msg
    end function
    integer(c_int) function fv3_halo_message_elems(ctx, nfields, fields, elems) bind(C, name="fv3_halo_message_elems")
      import :: c_int, c_ptr, c_size_t, fv3_halo_field
      type(c_ptr), value :: ctx
      integer(c_int), value :: nfields
      type(fv3_halo_field), intent(in) :: fields(*)
      integer(c_size_t), intent(out) :: elems(8)
    end function
    integer(c_int) function fv3_halo_periodic_group(ctx, nfields, fields) bind(C, name="fv3_halo_periodic_group")
      import :: c_int, c_ptr, fv3_halo_field
      type(c_ptr), value :: ctx
      integer(c_int), value :: nfields
      type(fv3_halo_field), intent(in) :: fields(*)
    end function
    integer(c_int) function fv3_halo_pack(ctx, nfields, fields, sendbuf) bind(C, name="fv3_halo_pack")
      import :: c_int, c_ptr, fv3_halo_field
      type(c_ptr), value :: ctx
      integer(c_int), value :: nfields
      type(fv3_halo_field), intent(in) :: fields(*)
      type(c_ptr), intent(in) :: sendbuf(8)
    end function
    integer(c_int) function fv3_halo_unpack(ctx, nfields, fields, recvbuf) bind(C, name="fv3_halo_unpack")
      import :: c_int, c_ptr, fv3_halo_field
      type(c_ptr), value :: ctx
      integer(c_int), value :: nfields
      type(fv3_halo_field), intent(in) :: fields(*)
      type(c_ptr), intent(in) :: recvbuf(8)
    end function
    ! the exchange behind the C ABI (RCCL on a stream the context owns): start / complete_group_halo_update, mp_reduce_max
    integer(c_int) function fv3_comm_get_unique_id(id) bind(C, name="fv3_comm_get_unique_id")
      import :: c_int, c_signed_char
      integer(c_signed_char), intent(out) :: id(128)
    end function
    integer(c_int) function fv3_comm_init(ctx, rank, nranks, id) bind(C, name="fv3_comm_init")
      import :: c_int, c_ptr, c_signed_char
      type(c_ptr), value :: ctx
      integer(c_int), value :: rank, nranks
      integer(c_signed_char), intent(in) :: id(128)
    end function
    integer(c_int) function fv3_comm_destroy(ctx) bind(C, name="fv3_comm_destroy")
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
    end function
    integer(c_int) function fv3_halo_start(ctx, nfields, fields, to, from) bind(C, name="fv3_halo_start")
      import :: c_int, c_ptr, fv3_halo_field
      type(c_ptr), value :: ctx
      integer(c_int), value :: nfields
      type(fv3_halo_field), intent(in) :: fields(*)
      integer(c_int), intent(in) :: to(8), from(8)
    end function
    integer(c_int) function fv3_halo_complete(ctx) bind(C, name="fv3_halo_complete")
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
    end function
    ! the cube-edge exchange (mpp_update_domains / mpp_get_boundary on the cubed-sphere mosaic, tools/fv_mp_mod.F90:498-546)
    integer(c_long) function fv3_cube_table(npx, ng, kind, member, face, dst, src_face, comp, src, sgn) bind(C, name="fv3_cube_table")
      import :: c_int, c_long, c_ptr
      integer(c_int), value :: npx, ng, kind, member, face
      type(c_ptr), value :: dst, src_face, comp, src, sgn
    end function
    integer(c_int) function fv3_cube_halo_start(nctx, ctxs, faces, face_rank, nfields, fields) bind(C, name="fv3_cube_halo_start")
      import :: c_int, c_ptr, fv3_cube_field
      integer(c_int), value :: nctx, nfields
      type(c_ptr), intent(in) :: ctxs(*)
      integer(c_int), intent(in) :: faces(*), face_rank(6)
      type(fv3_cube_field), intent(in) :: fields(*)
    end function
    integer(c_int) function fv3_cube_halo_complete(nctx, ctxs) bind(C, name="fv3_cube_halo_complete")
      import :: c_int, c_ptr
      integer(c_int), value :: nctx
      type(c_ptr), intent(in) :: ctxs(*)
    end function
    integer(c_int) function fv3_allreduce_max(ctx, buf, n) bind(C, name="fv3_allreduce_max")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx
      real(c_double), intent(inout) :: buf(*)
      integer(c_int), value :: n
    end function
    integer(c_int) function fv3_omga_update(ctx, rdt, ptop, pe, delp_before, omga) bind(C, name="fv3_omga_update")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx, pe, delp_before, omga
      real(c_double), value :: rdt, ptop
    end function
    integer(c_int) function fv3_pt_to_theta_v(ctx, hydrostatic, zvir, kappa, rdgas, grav, pt, delp, delz, qv, pkz) &
        bind(C, name="fv3_pt_to_theta_v")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx, pt, delp, delz, qv, pkz
      integer(c_int), value :: hydrostatic
      real(c_double), value :: zvir, kappa, rdgas, grav
    end function
    integer(c_int) function fv3_c2l(ctx, c2l_ord, u, v, ua, va) bind(C, name="fv3_c2l")
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx, u, v, ua, va
      integer(c_int), value :: c2l_ord
    end function
    integer(c_int) function fv3_rayleigh_u2f(ctx, kmax, hydrostatic, u, v, w, ua, va, u2f) bind(C, name="fv3_rayleigh_u2f")
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx, u, v, w, ua, va, u2f
      integer(c_int), value :: kmax, hydrostatic
    end function
    integer(c_int) function fv3_rayleigh_apply(ctx, kmax, conserve, hydrostatic, cp, rg, ptop, pm, rf, u2f, pt, delz, &
                                               u, v, w) bind(C, name="fv3_rayleigh_apply")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx, u2f, pt, delz, u, v, w
      integer(c_int), value :: kmax, conserve, hydrostatic
      real(c_double), value :: cp, rg, ptop
      real(c_double), intent(in) :: pm(*), rf(*)
    end function
    integer(c_int) function fv3_rayleigh_super(ctx, kmax, conserve, hydrostatic, cp, rg, ptop, pm, rf, ua, va, pt, &
                                               u, v, w, u00, v00) bind(C, name="fv3_rayleigh_super")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx, ua, va, pt, u, v, w, u00, v00
      integer(c_int), value :: kmax, conserve, hydrostatic
      real(c_double), value :: cp, rg, ptop
      real(c_double), intent(in) :: pm(*), rf(*)
    end function
    integer(c_int) function fv3_adv_pe(ctx, ptop, ua, va, delp_before, omga) bind(C, name="fv3_adv_pe")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx, ua, va, delp_before, omga
      real(c_double), value :: ptop
    end function
    integer(c_int) function fv3_ordered_sum(ctx, values, n, total) bind(C, name="fv3_ordered_sum")
      import :: c_int, c_ptr, c_double, c_size_t
      type(c_ptr), value :: ctx
      real(c_double), intent(in) :: values(*)
      integer(c_size_t), value :: n
      real(c_double), intent(out) :: total
    end function
    integer(c_int) function fv3_compute_total_energy(ctx, p, moist_phys, u, v, w, delz, pt, delp, q, qc, pe, peln, phis, &
                                                     te_2d) bind(C, name="fv3_compute_total_energy")
      import :: c_int, c_ptr, fv3_remap_params
      type(c_ptr), value :: ctx, u, v, w, delz, pt, delp, q, qc, pe, peln, phis, te_2d
      type(fv3_remap_params), intent(in) :: p
      integer(c_int), value :: moist_phys
    end function
    integer(c_int) function fv3_energy_fixer_sums(ctx, p, only_sums, u, v, w, delz, pt, delp, q, pe, peln, phis, pkz, pk, &
                                                  te0_2d, te_2d, zsum1, zsum0) bind(C, name="fv3_energy_fixer_sums")
      import :: c_int, c_ptr, fv3_remap_params
      type(c_ptr), value :: ctx, u, v, w, delz, pt, delp, q, pe, peln, phis, pkz, pk, te0_2d, te_2d, zsum1, zsum0
      type(fv3_remap_params), intent(in) :: p
      integer(c_int), value :: only_sums
    end function
    integer(c_int) function fv3_remap_finish(ctx, p, dtmp, pt, pkz, q) bind(C, name="fv3_remap_finish")
      import :: c_int, c_ptr, c_double, fv3_remap_params
      type(c_ptr), value :: ctx, pt, pkz, q
      type(fv3_remap_params), intent(in) :: p
      real(c_double), value :: dtmp
    end function
    integer(c_int) function fv3_divg2_ext(ctx, d_ext, delp, vt, divg2) bind(C, name="fv3_divg2_ext")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx, delp, vt, divg2
      real(c_double), value :: d_ext
    end function
    integer(c_int) function fv3_one_grad_p(ctx, u, v, pk, gz, divg2, dt, ptk) bind(C, name="fv3_one_grad_p")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx, u, v, pk, gz, divg2
      real(c_double), value :: dt, ptk
    end function
    integer(c_int) function fv3_one_grad_p_nh(ctx, u, v, pk, gz, divg2, delp, dt, ptop, gz_scale) bind(C, name="fv3_one_grad_p_nh")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx, u, v, pk, gz, divg2, delp
      real(c_double), value :: dt, ptop, gz_scale
    end function
    integer(c_int) function fv3_grad1_p_update(ctx, divg2, u, v, pk, gz, dt, ptk, beta, du, dv) bind(C, name="fv3_grad1_p_update")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx, divg2, u, v, pk, gz, du, dv
      real(c_double), value :: dt, ptk, beta
    end function
    integer(c_int) function fv3_split_p_grad(ctx, u, v, pp, gz, gz_scale, delp, pk, beta, dt, top_value, du, dv) &
        bind(C, name="fv3_split_p_grad")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx, u, v, pp, gz, delp, pk, du, dv
      real(c_double), value :: gz_scale, beta, dt, top_value
    end function
    integer(c_int) function fv3_profile_report_timers(ctx, out, cap) bind(C, name="fv3_profile_report_timers")
      import :: c_int, c_ptr, c_char, c_size_t
      type(c_ptr), value :: ctx
      character(kind=c_char), intent(inout) :: out(*)
      integer(c_size_t), value :: cap
    end function
    integer(c_int) function fv3_prt_maxmin(ctx, q, nk, fac, out) bind(C, name="fv3_prt_maxmin")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx, q
      integer(c_int), value :: nk
      real(c_double), value :: fac
      real(c_double), intent(out) :: out(3)
    end function
    integer(c_int) function fv3_set_remap_te(ctx, remap_te, hs, te) bind(C, name="fv3_set_remap_te")
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx, hs, te
      integer(c_int), value :: remap_te
    end function
    integer(c_int) function fv3_d_sw_inline_q(ctx, nq, hord_tr, nord_t, damp_t, q, q_out, delp_old, delp_new, fx, fy, crx, cry, &
                                              xfx, yfx) bind(C, name="fv3_d_sw_inline_q")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx, q, q_out, delp_old, delp_new, fx, fy, crx, cry, xfx, yfx
      integer(c_int), value :: nq, hord_tr, nord_t
      real(c_double), value :: damp_t
    end function
    integer(c_int) function fv3_flux_accum(ctx, mfx, mfy, fx, fy) bind(C, name="fv3_flux_accum")
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx, mfx, mfy, fx, fy
    end function
    integer(c_int) function fv3_fill2d_mass(ctx, nk, q, delp, qt) bind(C, name="fv3_fill2d_mass")
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx, q, delp, qt
      integer(c_int), value :: nk
    end function
    integer(c_int) function fv3_fill2d_apply(ctx, nk, qt, delp, q) bind(C, name="fv3_fill2d_apply")
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx, qt, delp, q
      integer(c_int), value :: nk
    end function
    integer(c_int) function fv3_copy_a_to_cc(ctx, src, dst, nk) bind(C, name="fv3_copy_a_to_cc")
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx, src, dst
      integer(c_int), value :: nk
    end function
    integer(c_int) function fv3_heat_source_accum(ctx, heat_source, heat_s) bind(C, name="fv3_heat_source_accum")
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx, heat_source, heat_s
    end function
    integer(c_int) function fv3_del2_cubed(ctx, q, nk, cd, nmax) bind(C, name="fv3_del2_cubed")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx, q
      integer(c_int), value :: nk, nmax
      real(c_double), value :: cd
    end function
    integer(c_int) function fv3_apply_heat_source(ctx, n_con, hydrostatic, bdt, delt_max, cp_air, cv_air, rdgas, grav, &
                                                  pt, heat_source, delp, delz, pkz) bind(C, name="fv3_apply_heat_source")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx, pt, heat_source, delp, delz, pkz
      integer(c_int), value :: n_con, hydrostatic
      real(c_double), value :: bdt, delt_max, cp_air, cv_air, rdgas, grav
    end function
    ! ---- nonhydrostatic column path, vertical remap, tracer transport --------------------------------
    integer(c_int) function fv3_memcpy_d2d(ctx, dst, src, bytes) bind(C, name="fv3_memcpy_d2d")
      import :: c_int, c_ptr, c_size_t
      type(c_ptr), value :: ctx, dst, src
      integer(c_size_t), value :: bytes
    end function
    integer(c_int) function fv3_set_dp_ref(ctx, dp0) bind(C, name="fv3_set_dp_ref")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx
      real(c_double), intent(in) :: dp0(*)          ! host, npz
    end function
    integer(c_int) function fv3_update_dz_c(ctx, dt, zs, ut, vt, gz_in, gz, ws) bind(C, name="fv3_update_dz_c")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx, zs, ut, vt, gz_in, gz, ws
      real(c_double), value :: dt
    end function
    integer(c_int) function fv3_set_condensate(ctx, q_con, cappa) bind(C, name="fv3_set_condensate")
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx, q_con, cappa
    end function
    integer(c_int) function fv3_registry_mode(ctx, lazy) bind(C, name="fv3_registry_mode")   ! host-address field registry (include/fv3_mi355x.h)
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
      integer(c_int), value :: lazy
    end function
    integer(c_int) function fv3_registry_put(ctx, dev, host, bytes) bind(C, name="fv3_registry_put")
      import :: c_int, c_ptr, c_size_t
      type(c_ptr), value :: ctx, dev, host
      integer(c_size_t), value :: bytes
    end function
    integer(c_int) function fv3_registry_get(ctx, host, dev, bytes) bind(C, name="fv3_registry_get")
      import :: c_int, c_ptr, c_size_t
      type(c_ptr), value :: ctx, host, dev
      integer(c_size_t), value :: bytes
    end function
    integer(c_int) function fv3_registry_host_touched(ctx, host) bind(C, name="fv3_registry_host_touched")
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx, host
    end function
    integer(c_int) function fv3_registry_fetch(ctx, host) bind(C, name="fv3_registry_fetch")
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx, host
    end function
    !> an array that is deallocated or rebound leaves the registry (host = c_null_ptr: every entry); discard = 0 fetches it first
    integer(c_int) function fv3_registry_forget(ctx, host, discard) bind(C, name="fv3_registry_forget")
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx, host
      integer(c_int), value :: discard
    end function
    integer(c_int) function fv3_registry_stats(ctx, out4) bind(C, name="fv3_registry_stats")
      import :: c_int, c_ptr, c_long_long
      type(c_ptr), value :: ctx
      integer(c_long_long), intent(out) :: out4(4)
    end function
    !> replaces: if (flagstruct%fill_dp) call mix_dp(hydrostatic, w, delp, pt, npz, ak, bk, .false., fv_debug, bd, gridstruct)   dyn_core.F90:820
    integer(c_int) function fv3_mix_dp(ctx, hydrostatic, w, delp, pt) bind(C, name="fv3_mix_dp")
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx, w, delp, pt
      integer(c_int), value :: hydrostatic
    end function
    integer(c_int) function fv3_compute_aam(ctx, radius, omega, agrav, ptop, coslat, ua, delp, aam, m_fac, ps) bind(C, name="fv3_compute_aam")   ! consv_am
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx, coslat, ua, delp, aam, m_fac, ps
      real(c_double), value :: radius, omega, agrav, ptop
    end function
    integer(c_int) function fv3_consv_am_apply(ctx, u00, l2c_u, l2c_v, u, v) bind(C, name="fv3_consv_am_apply")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx, l2c_u, l2c_v, u, v
      real(c_double), value :: u00
    end function
    integer(c_int) function fv3_set_fast_tau_w(ctx, k_rf, rff) bind(C, name="fv3_set_fast_tau_w")   ! fast_tau_w_sec > 0: rff(1:k_rf), host
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx
      integer(c_int), value :: k_rf
      real(c_double), intent(in) :: rff(*)
    end function
    integer(c_int) function fv3_set_ray_fast(ctx, kmax, k_rf, dm, rf, dp) bind(C, name="fv3_set_ray_fast")   ! Ray_fast's first call
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx
      integer(c_int), value :: kmax, k_rf
      real(c_double), value :: dm
      real(c_double), intent(in) :: rf(*), dp(*)
    end function
    integer(c_int) function fv3_ray_fast(ctx, u, v, w, hydrostatic) bind(C, name="fv3_ray_fast")   ! dyn_core.F90:1057-1060
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx, u, v, w
      integer(c_int), value :: hydrostatic
    end function
    integer(c_int) function fv3_riem_solver_c(ctx, dt, cn, hs, w3, pt, delp, gz, pef, ws) &
        bind(C, name="fv3_riem_solver_c")
      import :: c_int, c_ptr, c_double, fv3_nh_consts
      type(c_ptr), value :: ctx, hs, w3, pt, delp, gz, pef, ws
      real(c_double), value :: dt
      type(fv3_nh_consts), intent(in) :: cn
    end function
    integer(c_int) function fv3_update_dz_d(ctx, hord, zs, zh_in, zh_out, crx, cry, xfx, yfx, ws, rdt) &
        bind(C, name="fv3_update_dz_d")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx, zs, zh_in, zh_out, crx, cry, xfx, yfx, ws
      integer(c_int), value :: hord
      real(c_double), value :: rdt
    end function
    integer(c_int) function fv3_riem_solver3(ctx, dt, cn, zs, w, delz, pt, delp, zh, pe, ppe, pk3, pk, peln, ws, &
                                             use_logp, last_call, fp_out) bind(C, name="fv3_riem_solver3")
      import :: c_int, c_ptr, c_double, fv3_nh_consts
      type(c_ptr), value :: ctx, zs, w, delz, pt, delp, zh, pe, ppe, pk3, pk, peln, ws
      real(c_double), value :: dt
      type(fv3_nh_consts), intent(in) :: cn
      integer(c_int), value :: use_logp, last_call, fp_out
    end function
    integer(c_int) function fv3_p_grad_c(ctx, dt2, delpc, pkc, gz, uc, vc, hydrostatic) bind(C, name="fv3_p_grad_c")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx, delpc, pkc, gz, uc, vc
      real(c_double), value :: dt2
      integer(c_int), value :: hydrostatic
    end function
    integer(c_int) function fv3_nh_p_grad(ctx, u, v, pp, gz, gz_scale, delp, pk, dt, top_value) &
        bind(C, name="fv3_nh_p_grad")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx, u, v, pp, gz, delp, pk
      real(c_double), value :: gz_scale, dt, top_value
    end function
    integer(c_int) function fv3_zh_from_delz(ctx, zs, delz, zh) bind(C, name="fv3_zh_from_delz")
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx, zs, delz, zh
    end function
    integer(c_int) function fv3_pk3_halo(ctx, ptop, akap, pk3, delp, use_logp) bind(C, name="fv3_pk3_halo")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx, pk3, delp
      real(c_double), value :: ptop, akap
      integer(c_int), value :: use_logp
    end function
    integer(c_int) function fv3_pe_halo(ctx, ptop, pe, delp) bind(C, name="fv3_pe_halo")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx, pe, delp
      real(c_double), value :: ptop
    end function
    integer(c_int) function fv3_geopk(ctx, ptop, akap, cp_air, ptk, pe, peln, delp, pk, gz, hs, pt, pkz, cg) &
        bind(C, name="fv3_geopk")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx, pe, peln, delp, pk, gz, hs, pt, pkz
      real(c_double), value :: ptop, akap, cp_air, ptk
      integer(c_int), value :: cg
    end function
    integer(c_int) function fv3_set_moist(ctx, m, q_con, cappa) bind(C, name="fv3_set_moist")
      import :: c_int, c_ptr, fv3_moist_params
      type(c_ptr), value :: ctx, q_con, cappa
      type(fv3_moist_params), intent(in) :: m
    end function
    integer(c_int) function fv3_set_ak_bk(ctx, ak, bk) bind(C, name="fv3_set_ak_bk")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx
      real(c_double), intent(in) :: ak(*), bk(*)    ! host, npz+1
    end function
    integer(c_int) function fv3_lagrangian_to_eulerian(ctx, p, kord_tr, ps, pe, delp, pkz, pk, u, v, w, delz, pt, q, &
                                                       peln, omga, ws) bind(C, name="fv3_lagrangian_to_eulerian")
      import :: c_int, c_ptr, fv3_remap_params
      type(c_ptr), value :: ctx, ps, pe, delp, pkz, pk, u, v, w, delz, pt, q, peln, omga, ws
      type(fv3_remap_params), intent(in) :: p
      integer(c_int), intent(in) :: kord_tr(*)      ! host, nq
    end function
    integer(c_int) function fv3_tracer_2d_prep(ctx, q_split, cx, cy, xfx, yfx, cmax) bind(C, name="fv3_tracer_2d_prep")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx, cx, cy, xfx, yfx
      integer(c_int), value :: q_split
      real(c_double), intent(out) :: cmax(*)        ! host, npz
    end function
    integer(c_int) function fv3_tracer_2d_scale(ctx, frac, cx, xfx, mfx, cy, yfx, mfy) bind(C, name="fv3_tracer_2d_scale")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx, cx, xfx, mfx, cy, yfx, mfy
      real(c_double), intent(in) :: frac(*)         ! host, npz
    end function
    integer(c_int) function fv3_tracer_2d_step(ctx, it, nsplt, ksplt, nq, hord, nord_tr, trdm, q, q_out, dp1, dp1_out, &
                                               mfx, mfy, cx, cy, xfx, yfx) bind(C, name="fv3_tracer_2d_step")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx, q, q_out, dp1, dp1_out, mfx, mfy, cx, cy, xfx, yfx
      integer(c_int), value :: it, nsplt, nq, hord, nord_tr
      integer(c_int), intent(in) :: ksplt(*)        ! host, npz
      real(c_double), value :: trdm
    end function
  end interface

contains

  !> the reference aborts through mpp_error(FATAL); the C ABI returns a status instead
  subroutine fv3_check(rc, where)
    integer(c_int), intent(in) :: rc
    character(len=*), intent(in) :: where
    if (rc /= 0) then
      write(*,*) 'FATAL in ', where, ': fv3_mi355x status ', rc
      error stop 1
    end if
  end subroutine

end module fv3_mi355x_mod
