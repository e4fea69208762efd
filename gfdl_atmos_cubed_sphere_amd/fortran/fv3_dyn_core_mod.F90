!> dyn_core with the REFERENCE'S argument list (model/dyn_core.F90:94-98), over the device-resident substep loop of
!> fv3_host_mod: the form a model that keeps its state in host arrays with the fv_arrays layout calls without touching
!> its own code.  Every array argument has the reference's bounds (fv_arrays.F90:1521-1563), the scalars and the derived
!> types the reference's names.
!>
!> The derived types of the interface -- fv_grid_bounds_type, fv_grid_type, fv_flags_type, fv_nest_type, fv_thermo_type,
!> fv_diag_type (fv_arrays_mod), domain2d (mpp_domains_mod), group_halo_update_type (fv_mp_mod) -- are declared in
!> fv3_arrays_compat_mod below with the members THIS PATH READS under the reference's member names (fv_arrays.F90:75-205,
!> :207-906, :1192-1200).  fv_arrays_mod itself cannot be compiled without FMS; inside the model a maintainer replaces
!> `use fv3_arrays_compat_mod` by `use fv_arrays_mod` / `use fv_mp_mod` for the fv_arrays types.  domain2d is opaque in FMS: this file
!> reads it ONLY through the accessors fv3_domain_pe / _npes / _tile / _layout / _tile_pe / _comm_id of the compat module, whose bodies
!> a maintainer re-points at mpp_pe, mpp_npes, mpp_get_tile_id, mpp_get_layout, mpp_get_tile_pelist (the call sites stay); the tracer
!> indices come through get_tracer_index(MODEL_ATMOS, ...) as in fv_dynamics.F90:275-283 (tracer_manager_mod's in the model).
!>
!> What a call does: (first call) creates the context from bd / gridstruct / flagstruct and uploads the metric terms once;
!> (every call) host -> device copies of the prognostic arrays, the substep loop (fv3_dyn_core: n_split substeps, the d_con
!> heating; hydrostatic or not), device -> host copies of everything dyn_core leaves to its caller.  The copies cross PCIe
!> (≈ 63 GB/s): this is the compatibility form; the resident form (fv3_host_mod's own fv3_fv_dynamics, which keeps the
!> state on the device across dyn_core, tracer_2d and the remap) is the fast one.
!>
!> grid_type < 3 (round 4): a tile of the CUBED SPHERE per call -- one context per tile this process holds, the cubed members of
!> gridstruct, domain -> which tile / which PE holds which tile; the tiles of a process are given one call after the other and the
!> call of the last one runs the step for all of them (see fv_dynamics_sphere); with several processes the exchange runs between
!> them.  fv_dynamics there carries consv_te (compute_total_energy + the energy fixer), tau > 0 (Rayleigh_Super) and the virtual
!> effect in Fortran.
!>
!> grid_type = 4 on several PEs (round 4): domain%layout, %pe, %npes, %comm_id -- the group halo updates through fv3_halo_start /
!> fv3_halo_complete with the neighbour PEs, tracer_2d's mp_reduce_max through fv3_allreduce_max (fv3_host_comm_layout).
!>
!> Restrictions (error stop with the reason, never a silent difference): no nesting / regional BCs;
!> hybrid_z has no effect on this path in the reference either (fv_mapz.F90:62, :128); beta < 0 only as beta < -0.1 in a nonhydrostatic run.  consv_am (gridstruct%agrid, %l2c_u, %l2c_v, idiag%zxg),
!> do_diss_est (the SKEB diss_est accumulation), fill_dp (mix_dp), consv_te, tau > 0, RF_fast, fast_tau_w_sec and
!> thermostruct%use_cond / moist_kappa (the reference's defaults) are carried on both domains.
module fv3_arrays_compat_mod
  use iso_c_binding
  implicit none
  public

  type fv_grid_bounds_type                    ! fv_arrays.F90:1192-1200
    integer :: is, ie, js, je
    integer :: isd, ied, jsd, jed
    integer :: isc, iec, jsc, jec
    integer :: ng = 3
  end type

  type fv_grid_type                           ! fv_arrays.F90:75-205; shapes :1749-1881
    real(c_double), allocatable, dimension(:,:) :: area, rarea, dxa, dya, rdxa, rdya, cosa_s, rsin2, f0       ! (isd:ied, jsd:jed)
    real(c_double), allocatable, dimension(:,:) :: dx, rdx, dyc, rdyc, cosa_v, sina_v, rsin_v, divg_u, del6_u ! (isd:ied, jsd:jed+1)
    real(c_double), allocatable, dimension(:,:) :: dy, rdy, dxc, rdxc, cosa_u, sina_u, rsin_u, divg_v, del6_v ! (isd:ied+1, jsd:jed)
    real(c_double), allocatable, dimension(:,:) :: rarea_c, fC, cosa, sina                                    ! (isd:ied+1, jsd:jed+1)
    real(c_double), allocatable, dimension(:,:,:) :: sin_sg, cos_sg                                           ! (isd:ied, jsd:jed, 9)
    real(c_double) :: da_min = 0.d0, da_min_c = 0.d0
    integer :: grid_type = 4
    logical :: nested = .false., bounded_domain = .false., regional = .false., stretched_grid = .false.
    ! the members a cubed-sphere tile adds (grid_type < 3), with the reference's shapes (fv_arrays.F90:1778-1854)
    real(c_double), allocatable :: grid(:,:,:), agrid(:,:,:)                 ! (isd:ied+1, jsd:jed+1, 2), (isd:ied, jsd:jed, 2): lon, lat
    real(c_double), allocatable :: l2c_u(:,:), l2c_v(:,:)                    ! (is:ie, js:je+1), (is:ie+1, js:je): fv_arrays.F90:112, :1812-1813 (consv_am)
    real(c_double), allocatable :: edge_s(:), edge_n(:), edge_w(:), edge_e(:)   ! (npx), (npx), (npy), (npy)
    real(c_double), allocatable :: rsina(:,:)                                ! (is:ie+1, js:je+1)
    real(c_double), allocatable, dimension(:,:) :: a11, a12, a21, a22        ! (is-1:ie+1, js-1:je+1)
    real(c_double), allocatable, dimension(:,:,:) :: ec1, ec2                ! (3, isd:ied, jsd:jed)
    real(c_double), allocatable, dimension(:,:,:) :: en1, en2                ! (3, is:ie, js:je+1), (3, is:ie+1, js:je)
    ! NOT a member of the reference's type: the extrap_corner factors of a2b_ord4, which the reference forms from grid / agrid at
    ! every call (a2b_edge.F90:83-112, :452-462).  Left at its default they are formed from grid / agrid here, once.
    real(c_double) :: corner_f(12) = -1.d0
  end type

  type fv_flags_type                          ! fv_arrays.F90:207-906 (the members the substep loop reads; the reference's defaults)
    integer :: grid_type = 0
    integer :: n_split = 0, k_split = 1, q_split = 0
    integer :: nord = 1, nord_tr = 0
    real(c_double) :: d4_bg = 0.16d0, d2_bg = 0.d0, d2_bg_k1 = 4.d0, d2_bg_k2 = 2.d0
    real(c_double) :: dddmp = 0.d0, vtdm4 = 0.d0, d_con = 0.d0, ke_bg = 0.d0, trdm2 = 0.d0
    real(c_double) :: d_ext = 0.02d0, delt_max = 1.d0, beta = 0.d0, lim_fac = 1.d0
    real(c_double) :: a_imp = 0.75d0, p_fac = 0.05d0
    integer :: m_split = 0
    integer :: n_sponge = 1
    integer :: hord_mt = 10, hord_vt = 10, hord_tm = 10, hord_dp = 10, hord_tr = 8
    integer :: kord_tm = -8, kord_mt = 8, kord_wz = 8, kord_tr = 8
    logical :: do_vort_damp = .false., use_logp = .false., use_old_omega = .true., is_ideal_case = .false.
    logical :: convert_ke = .false., hydrostatic = .true., adiabatic = .false., fill = .false.
    logical :: do_diss_est = .false., prevent_diss_cooling = .false., do_f3d = .false., inline_q = .false., fill_dp = .false.
    logical :: nested = .false., regional = .false.
    real(c_double) :: tau = 0.d0, rf_cutoff = 30.d2, fast_tau_w_sec = 0.d0
    logical :: RF_fast = .false., consv_am = .false., do_sat_adj = .false., moist_phys = .true.
    integer :: c2l_ord = 4, nwat = 3
  end type

  type fv_nest_type
    logical :: nested = .false.
  end type

  type fv_thermo_type
    logical :: use_cond = .false., moist_kappa = .false.
  end type

  type fv_diag_type
    integer :: id_divg = 0, id_ws = 0
    real(c_double), allocatable :: zxg(:,:)                                  ! (isc:iec, jsc:jec): fv_arrays.F90:60 (consv_am's mountain torque term)
  end type

  !> mpp_domains_mod's domain2d is opaque to the dynamical core; what this path needs of it is what mpp_define_mosaic was given
  !> (tools/fv_mp_mod.F90:498-546): which tile this call is for, which PE holds every tile, and -- with several PEs -- the
  !> id of the exchange's communicator (rank 0's fv3_comm_get_unique_id, distributed by the caller as it distributes anything else).
  type domain2d
    integer :: pe = 0, npes = 1
    integer :: tile = 1                        ! 1 .. 6
    integer :: face_rank(6) = 0                ! the PE that holds tile n
    integer :: layout(2) = 1                   ! grid_type = 4: the px x py layout of the doubly periodic domain (PE r holds block (mod(r, px), r / px))
    integer(c_signed_char) :: comm_id(128) = 0_c_signed_char
  end type

  type fv_atmos_type                          ! fv_arrays.F90:1270: only its presence in fv_dynamics' list (parent_grid) matters here
    integer :: grid_number = 1
  end type

  type inline_mp_type                         ! fv_arrays.F90: inline microphysics diagnostics (not on this path)
    integer :: unused = 0
  end type

  type group_halo_update_type                 ! fv_mp_mod.F90:646-876: the halo groups live behind fv3_halo_* / fv3_halo_fill_periodic
    integer :: id = 0
  end type

  !> tracer_manager_mod's get_tracer_index as fv_dynamics uses it (fv_dynamics.F90:275-283: sphum, liq_wat, ice_wat, rainwat, snowwat,
  !> graupel): FMS reads the indices from the field table; here the host registers them (fv3_register_tracer_index) and fv_dynamics asks
  !> with the reference's own call.  An unregistered tracer is NO_TRACER (< 0, as FMS's).
  integer, parameter :: MODEL_ATMOS = 1, NO_TRACER = 1 - huge(1)
  integer, parameter, private :: max_tracers_compat = 64
  character(len=32), private :: tracer_names_compat(max_tracers_compat) = ' '
  integer, private :: tracer_index_compat(max_tracers_compat) = NO_TRACER, n_tracers_compat = 0

contains

  subroutine fv3_register_tracer_index(name, index)
    character(len=*), intent(in) :: name
    integer, intent(in) :: index
    integer :: n
    do n = 1, n_tracers_compat
      if (trim(tracer_names_compat(n)) == trim(name)) then
        tracer_index_compat(n) = index
        return
      end if
    end do
    if (n_tracers_compat >= max_tracers_compat) error stop 'fv3_register_tracer_index: too many tracers'
    n_tracers_compat = n_tracers_compat + 1
    tracer_names_compat(n_tracers_compat) = name
    tracer_index_compat(n_tracers_compat) = index
  end subroutine

  integer function get_tracer_index(model, name)
    integer, intent(in) :: model
    character(len=*), intent(in) :: name
    integer :: n
    get_tracer_index = NO_TRACER
    if (model /= MODEL_ATMOS) return
    do n = 1, n_tracers_compat
      if (trim(tracer_names_compat(n)) == trim(name)) get_tracer_index = tracer_index_compat(n)
    end do
  end function

  !> What the dynamical core asks of a domain2d goes through accessors shaped like mpp_domains_mod's / mpp_mod's own (a maintainer's
  !> `use mpp_domains_mod` then replaces these bodies, not the call sites): the PE of this process, the number of PEs, the tile of this
  !> call, the io layout of the doubly periodic domain; and the two things FMS keeps inside the domain that this path needs as data --
  !> which PE holds every tile and the id of the exchange's communicator.
  integer function fv3_domain_pe(domain)
    type(domain2d), intent(in) :: domain
    fv3_domain_pe = domain%pe                  ! mpp_pe()
  end function
  integer function fv3_domain_npes(domain)
    type(domain2d), intent(in) :: domain
    fv3_domain_npes = domain%npes              ! mpp_npes()
  end function
  integer function fv3_domain_tile(domain)
    type(domain2d), intent(in) :: domain
    fv3_domain_tile = domain%tile              ! mpp_get_tile_id(domain)
  end function
  subroutine fv3_domain_layout(domain, layout)
    type(domain2d), intent(in) :: domain
    integer, intent(out) :: layout(2)
    layout = domain%layout                     ! mpp_get_layout(domain, layout)
  end subroutine
  integer function fv3_domain_tile_pe(domain, tile)
    type(domain2d), intent(in) :: domain
    integer, intent(in) :: tile
    fv3_domain_tile_pe = domain%face_rank(tile)   ! mpp_get_tile_pelist
  end function
  integer function fv3_domain_layout_of(domain, n)
    type(domain2d), intent(in) :: domain
    integer, intent(in) :: n
    integer :: layout(2)
    call fv3_domain_layout(domain, layout)
    fv3_domain_layout_of = layout(n)
  end function
  integer function fv3_tiles_held(domain, upto)   ! how many of the tiles 1 .. upto this PE holds
    type(domain2d), intent(in) :: domain
    integer, intent(in) :: upto
    integer :: t
    fv3_tiles_held = 0
    do t = 1, upto
      if (fv3_domain_tile_pe(domain, t) == fv3_domain_pe(domain)) fv3_tiles_held = fv3_tiles_held + 1
    end do
  end function
  function fv3_domain_tile_pelist(domain) result(pes)   ! the PE of every tile (mpp_get_tile_pelist, tile by tile)
    type(domain2d), intent(in) :: domain
    integer :: pes(6), t
    do t = 1, 6
      pes(t) = fv3_domain_tile_pe(domain, t)
    end do
  end function
  function fv3_domain_comm_id(domain) result(id)
    type(domain2d), intent(in) :: domain
    integer(c_signed_char) :: id(128)
    id = domain%comm_id
  end function
end module fv3_arrays_compat_mod


module fv3_dyn_core_mod
  use iso_c_binding
  use fv3_arrays_compat_mod
  use fv3_mi355x_mod
  use fv3_host_mod
  use fv3_sphere_mod
  implicit none
  private
  public :: dyn_core, dyn_core_end, fv_dynamics, fv_dynamics_end
  ! the host-address field registry of dyn_core on the doubly periodic domain (include/fv3_mi355x.h fv3_registry_*): lazy = the caller
  ! declares what it wrote between the calls (fv3_host_touched) and asks for what it reads (fv3_host_fetch); the rest stays on the device
  public :: fv3_dyn_core_registry, fv3_host_touched, fv3_host_fetch, fv3_dyn_core_registry_stats
  logical, save :: registry_lazy = .false.

  type(fv3_atmos), save :: at
  logical, save :: bound = .false.
  type(fv3_atmos), save :: atf          ! fv_dynamics keeps a context of its own (it carries the tracers)
  logical, save :: boundf = .false.
  ! the cubed sphere (grid_type < 3): one context per tile this process holds (fv3_sphere_mod); the host arrays of every tile's call
  type tile_arrays
    type(c_ptr) :: u, v, w, delz, pt, delp, q, ps, pe, pk, peln, pkz, omga, ua, va, uc, vc, mfx, mfy, cx, cy, q_con = c_null_ptr
    type(c_ptr) :: diss_est = c_null_ptr
  end type
  type(fv3_sphere), save :: sps
  type(tile_arrays), save :: tps(6)
  logical, save :: bound_s(6) = .false., comm_s = .false.
  type dc_tile_arrays
    type(c_ptr) :: u, v, w, delz, pt, delp, ws, pe, pk, peln, pkz, omga, ua, va, uc, vc, mfx, mfy, cx, cy, heat_source, q_con = c_null_ptr
    type(c_ptr) :: diss_est = c_null_ptr
  end type
  type(fv3_sphere), save :: spd         ! dyn_core's own contexts (no tracers), as on the doubly periodic domain
  type(dc_tile_arrays), save :: tpd(6)
  logical, save :: bound_d(6) = .false., comm_d = .false.

contains

  subroutine dyn_core(npx, npy, npz, ng, sphum, nq, bdt, n_map, n_split, zvir, cp, akap, cappa, grav, hydrostatic, &
                      u, v, w, delz, pt, q, delp, pe, pk, phis, ws, omga, ptop, pfull, ua, va, &
                      uc, vc, mfx, mfy, cx, cy, pkz, peln, q_con, ak, bk, &
                      ks, gridstruct, flagstruct, neststruct, thermostruct, idiag, bd, domain, &
                      init_step, i_pack, end_step, heat_source, diss_est, consv, te0_2d, time_total)
    integer, intent(in) :: npx, npy, npz, ng, nq, sphum, n_map, n_split, ks
    real(c_double), intent(in) :: bdt, zvir, cp, akap, grav, consv, ptop
    logical, intent(in) :: hydrostatic, init_step, end_step
    real(c_double), intent(in) :: pfull(npz), ak(npz+1), bk(npz+1)
    type(group_halo_update_type), intent(inout) :: i_pack(*)
    type(fv_grid_bounds_type), intent(in) :: bd
    real(c_double), intent(inout), target :: u(bd%isd:bd%ied, bd%jsd:bd%jed+1, npz)
    real(c_double), intent(inout), target :: v(bd%isd:bd%ied+1, bd%jsd:bd%jed, npz)
    real(c_double), intent(inout), target :: w(bd%isd:, bd%jsd:, 1:)
    real(c_double), intent(inout), target :: delz(bd%is:, bd%js:, 1:)
    real(c_double), intent(inout) :: cappa(bd%isd:, bd%jsd:, 1:)
    real(c_double), intent(inout), target :: pt(bd%isd:bd%ied, bd%jsd:bd%jed, npz), delp(bd%isd:bd%ied, bd%jsd:bd%jed, npz)
    real(c_double), intent(inout) :: q(bd%isd:bd%ied, bd%jsd:bd%jed, npz, nq)
    real(c_double), intent(inout), target :: heat_source(bd%isd:bd%ied, bd%jsd:bd%jed, npz)
    real(c_double), intent(inout), target :: diss_est(bd%isd:bd%ied, bd%jsd:bd%jed, npz)
    real(c_double), intent(in), optional :: time_total
    real(c_double), intent(inout), target :: phis(bd%isd:bd%ied, bd%jsd:bd%jed)
    real(c_double), intent(inout), target :: pe(bd%is-1:bd%ie+1, npz+1, bd%js-1:bd%je+1)
    real(c_double), intent(inout), target :: peln(bd%is:bd%ie, npz+1, bd%js:bd%je)
    real(c_double), intent(inout), target :: pk(bd%is:bd%ie, bd%js:bd%je, npz+1)
    real(c_double), intent(out), target :: ws(bd%is:bd%ie, bd%js:bd%je)
    real(c_double), intent(inout), target :: omga(bd%isd:bd%ied, bd%jsd:bd%jed, npz)
    real(c_double), intent(inout), target :: uc(bd%isd:bd%ied+1, bd%jsd:bd%jed, npz), vc(bd%isd:bd%ied, bd%jsd:bd%jed+1, npz)
    real(c_double), intent(inout), target, dimension(bd%isd:bd%ied, bd%jsd:bd%jed, npz) :: ua, va
    real(c_double), intent(inout), target :: q_con(bd%isd:, bd%jsd:, 1:)
    real(c_double), intent(inout) :: te0_2d(bd%is:bd%ie, bd%js:bd%je)
    real(c_double), intent(inout), target :: mfx(bd%is:bd%ie+1, bd%js:bd%je, npz), mfy(bd%is:bd%ie, bd%js:bd%je+1, npz)
    real(c_double), intent(inout), target :: cx(bd%is:bd%ie+1, bd%jsd:bd%jed, npz), cy(bd%isd:bd%ied, bd%js:bd%je+1, npz)
    real(c_double), intent(inout), target :: pkz(bd%is:bd%ie, bd%js:bd%je, npz)
    type(fv_grid_type), intent(inout), target :: gridstruct
    type(fv_flags_type), intent(in), target :: flagstruct
    type(fv_nest_type), intent(inout) :: neststruct
    type(fv_thermo_type), intent(inout), target :: thermostruct
    type(fv_diag_type), intent(in) :: idiag
    type(domain2d), intent(inout) :: domain

    real(c_double), allocatable, target :: w_c(:,:,:), delz_c(:,:,:), zs(:,:), qc_c(:,:,:), cp_c(:,:,:)
    logical :: moist
    integer(c_size_t) :: nk, nk1
    integer :: nx, ny

    if (neststruct%nested .or. gridstruct%nested .or. gridstruct%regional .or. gridstruct%bounded_domain) &
      error stop 'dyn_core (fv3_dyn_core_mod): nested / regional domains are not built'
    moist = thermostruct%use_cond .or. thermostruct%moist_kappa
    if (gridstruct%grid_type < 3) then
      call dyn_core_sphere()
      return
    end if
    if (gridstruct%grid_type /= 4) error stop 'dyn_core (fv3_dyn_core_mod): grid_type = 3 is not built'
    if (moist .and. hydrostatic) error stop 'dyn_core (fv3_dyn_core_mod): use_cond / moist_kappa are nonhydrostatic branches'
    if (thermostruct%use_cond .and. size(q_con, 3) < npz) error stop 'dyn_core (fv3_dyn_core_mod): use_cond needs q_con on npz levels'
    if (thermostruct%moist_kappa .and. size(cappa, 3) < npz) error stop 'dyn_core (fv3_dyn_core_mod): moist_kappa needs cappa on npz levels'
    if (flagstruct%beta < 0.d0 .and. (hydrostatic .or. flagstruct%beta >= -0.1d0)) &
      error stop 'dyn_core (fv3_dyn_core_mod): beta < 0: one_grad_p (beta < -0.1) is built for the nonhydrostatic loop'
    if (ng /= 3 .or. bd%ng /= 3) error stop 'dyn_core (fv3_dyn_core_mod): ng = 3'
    if (.not. bound) then
      call bind_context()
      if (fv3_domain_npes(domain) > 1) call fv3_host_comm_layout(at, fv3_domain_pe(domain), fv3_domain_npes(domain), fv3_domain_layout_of(domain, 1), fv3_domain_layout_of(domain, 2), fv3_domain_comm_id(domain))
    end if
    if (at%npz /= npz .or. at%is /= bd%is .or. at%ie /= bd%ie .or. at%js /= bd%js .or. at%je /= bd%je) &
      error stop 'dyn_core (fv3_dyn_core_mod): the domain changed between calls'
    at%fl%n_split = n_split
    nx = bd%ie - bd%is + 1; ny = bd%je - bd%js + 1
    nk = int(npz, c_size_t); nk1 = nk + 1
    call fv3_check(fv3_registry_mode(at%ctx, merge(1_c_int, 0_c_int, registry_lazy)), 'fv3_registry_mode')

    ! ---- host -> device: what the loop reads (dyn_core.F90:95-96: u, v, w, delz, pt, delp, phis; pkz / pe / pk / peln / omga /
    !      ua / va are intent(inout) members the loop only partly rewrites) ----
    call rput(at%u, c_loc(u), at%nU*nk);        call rput(at%v, c_loc(v), at%nV*nk)
    call rput(at%delp, c_loc(delp), at%nA*nk);  call rput(at%pt, c_loc(pt), at%nA*nk)
    call rput(at%phis, c_loc(phis), at%nA)
    allocate(zs(bd%isd:bd%ied, bd%jsd:bd%jed))
    zs = phis * (1.d0 / grav)                                              ! dyn_core.F90:246-251
    call put(at%zs, c_loc(zs), at%nA)
    if (.not. hydrostatic) then
      allocate(w_c(bd%isd:bd%ied, bd%jsd:bd%jed, npz), delz_c(bd%is:bd%ie, bd%js:bd%je, npz))   ! assumed-shape dummies: contiguous copies
      w_c = w(bd%isd:bd%ied, bd%jsd:bd%jed, 1:npz); delz_c = delz(bd%is:bd%ie, bd%js:bd%je, 1:npz)
      call put(at%w, c_loc(w_c), at%nA*nk);    call put(at%delz, c_loc(delz_c), at%nCC*nk)
    end if
    call rput(at%pkz, c_loc(pkz), at%nCC*nk);   call rput(at%pk, c_loc(pk), at%nCC*nk1)
    call rput(at%pe, c_loc(pe), int(nx+2, c_size_t)*nk1*(ny+2)); call rput(at%peln, c_loc(peln), at%nCC*nk1)
    call rput(at%omga, c_loc(omga), at%nA*nk);  call rput(at%ua, c_loc(ua), at%nA*nk); call rput(at%va, c_loc(va), at%nA*nk)
    ! thermostruct%use_cond: q_con (halo updated by the caller, fv_dynamics.F90:464) rides through d_sw and the Riemann solvers;
    ! moist_kappa: cappa (:465) is read by the solvers and the heating
    if (thermostruct%use_cond) then
      allocate(qc_c(bd%isd:bd%ied, bd%jsd:bd%jed, npz))
      qc_c = q_con(bd%isd:bd%ied, bd%jsd:bd%jed, 1:npz)
      call put(at%q_con, c_loc(qc_c), at%nA*nk)
    end if
    if (thermostruct%moist_kappa) then
      allocate(cp_c(bd%isd:bd%ied, bd%jsd:bd%jed, npz))
      cp_c = cappa(bd%isd:bd%ied, bd%jsd:bd%jed, 1:npz)
      call put(at%cappa, c_loc(cp_c), at%nA*nk)
    end if
    call fv3_check(fv3_sync(at%ctx), 'fv3_sync')

    if (.not. allocated(at%pfull)) at%pfull = pfull                        ! what Riem_Solver_c (:536) and Ray_fast (:1058) are handed
    if (flagstruct%do_diss_est) then                                       ! :285 (zero on init_step), then the caller's array rides along
      if (init_step) diss_est = 0.d0
      call diss_est_begin(at)
      call rput(at%diss_est, c_loc(diss_est), at%nA*nk)
    end if
    call fv3_dyn_core(at, bdt)                                             ! the substep loop (both branches), d_con heating

    ! ---- device -> host ----
    call fv3_check(fv3_sync(at%ctx), 'fv3_sync')
    call rget(c_loc(u), at%u, at%nU*nk);        call rget(c_loc(v), at%v, at%nV*nk)
    call rget(c_loc(delp), at%delp, at%nA*nk);  call rget(c_loc(pt), at%pt, at%nA*nk)
    if (.not. hydrostatic) then
      call get(c_loc(w_c), at%w, at%nA*nk);    call get(c_loc(delz_c), at%delz, at%nCC*nk)
      call rget(c_loc(ws), at%ws, at%nCC)
    end if
    call rget(c_loc(pkz), at%pkz, at%nCC*nk);   call rget(c_loc(pk), at%pk, at%nCC*nk1)
    call rget(c_loc(pe), at%pe, int(nx+2, c_size_t)*nk1*(ny+2)); call rget(c_loc(peln), at%peln, at%nCC*nk1)
    call rget(c_loc(omga), at%omga, at%nA*nk);  call rget(c_loc(ua), at%ua, at%nA*nk); call rget(c_loc(va), at%va, at%nA*nk)
    call rget(c_loc(uc), at%uc, at%nV*nk);      call rget(c_loc(vc), at%vc, at%nU*nk)
    call rget(c_loc(mfx), at%mfx, at%nFX*nk);   call rget(c_loc(mfy), at%mfy, at%nFY*nk)
    call rget(c_loc(cx), at%cx, at%nCX*nk);     call rget(c_loc(cy), at%cy, at%nCY*nk)
    if (flagstruct%d_con > 1.d-5) call rget(c_loc(heat_source), at%heat_source, at%nA*nk)
    if (flagstruct%do_diss_est) call rget(c_loc(diss_est), at%diss_est, at%nA*nk)
    if (thermostruct%use_cond) call get(c_loc(qc_c), at%q_con, at%nA*nk)
    call fv3_check(fv3_sync(at%ctx), 'fv3_sync')
    if (thermostruct%use_cond) q_con(bd%isd:bd%ied, bd%jsd:bd%jed, 1:npz) = qc_c
    if (.not. hydrostatic) then
      w(bd%isd:bd%ied, bd%jsd:bd%jed, 1:npz) = w_c; delz(bd%is:bd%ie, bd%js:bd%je, 1:npz) = delz_c
    else
      ws = 0.d0
    end if

  contains

    subroutine put(d, h, n)
      type(c_ptr), intent(in) :: d, h
      integer(c_size_t), intent(in) :: n
      call fv3_check(fv3_memcpy_h2d(at%ctx, d, h, n * 8_c_size_t), 'fv3_memcpy_h2d')
    end subroutine

    subroutine get(h, d, n)
      type(c_ptr), intent(in) :: d, h
      integer(c_size_t), intent(in) :: n
      call fv3_check(fv3_memcpy_d2h(at%ctx, h, d, n * 8_c_size_t), 'fv3_memcpy_d2h')
    end subroutine

    ! the caller's own arrays go through the registry: copied every time (eager) or only when the other copy is not current (lazy)
    subroutine rput(d, h, n)
      type(c_ptr), intent(in) :: d, h
      integer(c_size_t), intent(in) :: n
      call fv3_check(fv3_registry_put(at%ctx, d, h, n * 8_c_size_t), 'fv3_registry_put')
    end subroutine

    subroutine rget(h, d, n)
      type(c_ptr), intent(in) :: d, h
      integer(c_size_t), intent(in) :: n
      call fv3_check(fv3_registry_get(at%ctx, h, d, n * 8_c_size_t), 'fv3_registry_get')
    end subroutine

    !> grid_type < 3: a tile of the cubed sphere, as fv_dynamics below does it -- every call binds (first time) and uploads its tile,
    !> the call of the last tile this process holds runs the substep loop for all of them (fv3_sphere_dyn_core: every halo update
    !> through fv3_cube_halo_*, mpp_get_boundary after the last substep, adv_pe) and writes the results of every tile
    subroutine dyn_core_sphere()
      type(fv3_flags) :: fl
      integer :: slot, nloc, sl
      if (moist .and. hydrostatic) error stop 'dyn_core (fv3_dyn_core_mod): use_cond / moist_kappa are nonhydrostatic branches'
      if (thermostruct%use_cond) then     ! q_con is written back by the call of the LAST tile: the model's own array, whole and contiguous
        if (.not. is_contiguous(q_con) .or. size(q_con, 1) /= bd%ied - bd%isd + 1 .or. size(q_con, 2) /= bd%jed - bd%jsd + 1 .or. size(q_con, 3) /= npz) &
          error stop 'dyn_core (fv3_dyn_core_mod): use_cond on the cubed sphere needs q_con(isd:ied, jsd:jed, npz), contiguous'
      end if
      if (thermostruct%moist_kappa .and. size(cappa, 3) < npz) error stop 'dyn_core (fv3_dyn_core_mod): moist_kappa needs cappa on npz levels'
      if (ng /= 3 .or. bd%ng /= 3) error stop 'dyn_core (fv3_dyn_core_mod): ng = 3'
      if (fv3_domain_tile(domain) < 1 .or. fv3_domain_tile(domain) > 6) error stop 'dyn_core (fv3_dyn_core_mod): fv3_domain_tile(domain) must be 1 .. 6'
      if (fv3_domain_tile_pe(domain, fv3_domain_tile(domain)) /= fv3_domain_pe(domain)) error stop 'dyn_core (fv3_dyn_core_mod): this PE does not hold fv3_domain_tile(domain)'
      nloc = fv3_tiles_held(domain, 6)
      slot = fv3_tiles_held(domain, fv3_domain_tile(domain))
      nx = bd%ie - bd%is + 1; ny = bd%je - bd%js + 1
      nk = int(npz, c_size_t); nk1 = nk + 1
      if (.not. hydrostatic) then
        if (.not. (is_contiguous(w) .and. is_contiguous(delz)) .or. size(w, 1) /= nx + 6 .or. size(w, 2) /= ny + 6 .or. &
            size(w, 3) /= npz .or. size(delz, 1) /= nx .or. size(delz, 2) /= ny .or. size(delz, 3) /= npz) &
          error stop 'dyn_core (fv3_dyn_core_mod): w / delz must be the whole contiguous arrays of the tile'
      end if
      if (.not. bound_d(slot)) then
        call flags_of(flagstruct, fl)
        fl%ks = ks
        fl%n_split = n_split; fl%ptop = ptop; fl%grav = grav; fl%akap = akap; fl%cp_air = cp
        fl%hydrostatic = hydrostatic
        fl%use_cond = thermostruct%use_cond; fl%moist_kappa = thermostruct%moist_kappa
        fl%do_diss_est = flagstruct%do_diss_est
        call bind_sphere_tile(spd, slot, fv3_domain_tile(domain), npx, npy, npz, 0, bd, gridstruct, flagstruct, fl, ak, bk)
        bound_d(slot) = .true.
      end if
      associate (a => spd%f(slot))
        if (a%npz /= npz .or. a%ie /= bd%ie .or. a%je /= bd%je) error stop 'dyn_core (fv3_dyn_core_mod): the domain changed between calls'
        a%fl%n_split = n_split
        call puts(a, a%u, c_loc(u), a%nU*nk);        call puts(a, a%v, c_loc(v), a%nV*nk)
        call puts(a, a%delp, c_loc(delp), a%nA*nk);  call puts(a, a%pt, c_loc(pt), a%nA*nk)
        call puts(a, a%phis, c_loc(phis), a%nA)
        allocate(zs(bd%isd:bd%ied, bd%jsd:bd%jed))
        zs = phis * (1.d0 / grav)
        call puts(a, a%zs, c_loc(zs), a%nA)
        if (.not. hydrostatic) then
          call puts(a, a%w, c_loc(w), a%nA*nk);      call puts(a, a%delz, c_loc(delz), a%nCC*nk)
        end if
        call puts(a, a%pkz, c_loc(pkz), a%nCC*nk);   call puts(a, a%pk, c_loc(pk), a%nCC*nk1)
        call puts(a, a%pe, c_loc(pe), int(nx+2, c_size_t)*nk1*(ny+2)); call puts(a, a%peln, c_loc(peln), a%nCC*nk1)
        call puts(a, a%omga, c_loc(omga), a%nA*nk);  call puts(a, a%ua, c_loc(ua), a%nA*nk); call puts(a, a%va, c_loc(va), a%nA*nk)
        ! thermostruct%use_cond / moist_kappa: q_con, cappa of the tile (fv_dynamics updates their halos right after, :464-465: the
        ! cube-edge exchange in front of the substep loop does it here)
        if (thermostruct%use_cond) call puts(a, a%q_con, c_loc(q_con), a%nA*nk)
        if (thermostruct%moist_kappa) then
          allocate(cp_c(bd%isd:bd%ied, bd%jsd:bd%jed, npz))
          cp_c = cappa(bd%isd:bd%ied, bd%jsd:bd%jed, 1:npz)
          call puts(a, a%cappa, c_loc(cp_c), a%nA*nk)
        end if
        if (flagstruct%do_diss_est) then                                   ! :285 (zero on init_step), then the caller's array rides along
          if (init_step) diss_est = 0.d0
          call diss_est_begin(a)
          call puts(a, a%diss_est, c_loc(diss_est), a%nA*nk)
        end if
        call fv3_check(fv3_sync(a%ctx), 'fv3_sync')
      end associate
      tpd(slot)%diss_est = c_loc(diss_est)
      tpd(slot)%q_con = c_null_ptr
      if (thermostruct%use_cond) tpd(slot)%q_con = c_loc(q_con)
      tpd(slot)%u = c_loc(u); tpd(slot)%v = c_loc(v); tpd(slot)%pt = c_loc(pt); tpd(slot)%delp = c_loc(delp)
      tpd(slot)%w = c_null_ptr; tpd(slot)%delz = c_null_ptr
      if (.not. hydrostatic) then
        tpd(slot)%w = c_loc(w); tpd(slot)%delz = c_loc(delz)
      end if
      tpd(slot)%ws = c_loc(ws); tpd(slot)%pe = c_loc(pe); tpd(slot)%pk = c_loc(pk); tpd(slot)%peln = c_loc(peln)
      tpd(slot)%pkz = c_loc(pkz); tpd(slot)%omga = c_loc(omga); tpd(slot)%ua = c_loc(ua); tpd(slot)%va = c_loc(va)
      tpd(slot)%uc = c_loc(uc); tpd(slot)%vc = c_loc(vc); tpd(slot)%mfx = c_loc(mfx); tpd(slot)%mfy = c_loc(mfy)
      tpd(slot)%cx = c_loc(cx); tpd(slot)%cy = c_loc(cy); tpd(slot)%heat_source = c_loc(heat_source)
      if (slot < nloc) return                        ! the loop runs in the call of the last tile this process holds

      if (.not. comm_d) then
        if (fv3_domain_npes(domain) > 1) then
          call fv3_sphere_comm(spd, fv3_domain_pe(domain), fv3_domain_npes(domain), fv3_domain_tile_pelist(domain), fv3_domain_comm_id(domain))
        else
          call fv3_sphere_comm(spd, 0, 1, fv3_domain_tile_pelist(domain))
        end if
        comm_d = .true.
      end if
      if (moist) call fv3_sphere_halo_moist(spd)              ! what fv_dynamics does in front of dyn_core (:464-465 / :487-488)
      call fv3_sphere_dyn_core(spd, bdt, end_step)
      do sl = 1, nloc
        associate (a => spd%f(sl), tp => tpd(sl))
          call fv3_check(fv3_sync(a%ctx), 'fv3_sync')
          call gets(a, tp%u, a%u, a%nU*nk);        call gets(a, tp%v, a%v, a%nV*nk)
          call gets(a, tp%delp, a%delp, a%nA*nk);  call gets(a, tp%pt, a%pt, a%nA*nk)
          if (.not. hydrostatic) then
            call gets(a, tp%w, a%w, a%nA*nk);      call gets(a, tp%delz, a%delz, a%nCC*nk)
            call gets(a, tp%ws, a%ws, a%nCC)
          end if
          call gets(a, tp%pkz, a%pkz, a%nCC*nk);   call gets(a, tp%pk, a%pk, a%nCC*nk1)
          call gets(a, tp%pe, a%pe, int(nx+2, c_size_t)*nk1*(ny+2)); call gets(a, tp%peln, a%peln, a%nCC*nk1)
          call gets(a, tp%omga, a%omga, a%nA*nk);  call gets(a, tp%ua, a%ua, a%nA*nk); call gets(a, tp%va, a%va, a%nA*nk)
          call gets(a, tp%uc, a%uc, a%nV*nk);      call gets(a, tp%vc, a%vc, a%nU*nk)
          call gets(a, tp%mfx, a%mfx, a%nFX*nk);   call gets(a, tp%mfy, a%mfy, a%nFY*nk)
          call gets(a, tp%cx, a%cx, a%nCX*nk);     call gets(a, tp%cy, a%cy, a%nCY*nk)
          if (flagstruct%d_con > 1.d-5) call gets(a, tp%heat_source, a%heat_source, a%nA*nk)
          if (flagstruct%do_diss_est) call gets(a, tp%diss_est, a%diss_est, a%nA*nk)
          if (c_associated(tp%q_con)) call gets(a, tp%q_con, a%q_con, a%nA*nk)
          call fv3_check(fv3_sync(a%ctx), 'fv3_sync')
        end associate
      end do
    end subroutine

    subroutine puts(a, d, h, n)
      type(fv3_atmos), intent(in) :: a
      type(c_ptr), intent(in) :: d, h
      integer(c_size_t), intent(in) :: n
      call fv3_check(fv3_memcpy_h2d(a%ctx, d, h, n * 8_c_size_t), 'fv3_memcpy_h2d')
    end subroutine

    subroutine gets(a, h, d, n)
      type(fv3_atmos), intent(in) :: a
      type(c_ptr), intent(in) :: d, h
      integer(c_size_t), intent(in) :: n
      call fv3_check(fv3_memcpy_d2h(a%ctx, h, d, n * 8_c_size_t), 'fv3_memcpy_d2h')
    end subroutine

    !> first call: bounds + flags -> fv3_domain / fv3_flags, the gridstruct members by address -> fv3_grid_upload
    subroutine bind_context()
      type(fv3_domain) :: dom
      type(fv3_grid_host) :: gh
      type(fv3_flags) :: fl
      dom%is = bd%is; dom%ie = bd%ie; dom%js = bd%js; dom%je = bd%je; dom%ng = 3
      dom%npx = npx; dom%npy = npy; dom%npz = npz; dom%grid_type = gridstruct%grid_type
      dom%do_diss_est = merge(1, 0, flagstruct%do_diss_est); dom%prevent_diss_cooling = merge(1, 0, flagstruct%prevent_diss_cooling)
      dom%stretched_grid = merge(1, 0, gridstruct%stretched_grid); dom%lim_fac = flagstruct%lim_fac
      gh%da_min = gridstruct%da_min;        gh%da_min_c = gridstruct%da_min_c
      gh%area = c_loc(gridstruct%area);     gh%rarea = c_loc(gridstruct%rarea)
      gh%dxa = c_loc(gridstruct%dxa);       gh%dya = c_loc(gridstruct%dya)
      gh%rdxa = c_loc(gridstruct%rdxa);     gh%rdya = c_loc(gridstruct%rdya)
      gh%cosa_s = c_loc(gridstruct%cosa_s); gh%rsin2 = c_loc(gridstruct%rsin2);   gh%f0 = c_loc(gridstruct%f0)
      gh%dx = c_loc(gridstruct%dx);         gh%rdx = c_loc(gridstruct%rdx)
      gh%dyc = c_loc(gridstruct%dyc);       gh%rdyc = c_loc(gridstruct%rdyc)
      gh%cosa_v = c_loc(gridstruct%cosa_v); gh%sina_v = c_loc(gridstruct%sina_v); gh%rsin_v = c_loc(gridstruct%rsin_v)
      gh%divg_u = c_loc(gridstruct%divg_u); gh%del6_u = c_loc(gridstruct%del6_u)
      gh%dy = c_loc(gridstruct%dy);         gh%rdy = c_loc(gridstruct%rdy)
      gh%dxc = c_loc(gridstruct%dxc);       gh%rdxc = c_loc(gridstruct%rdxc)
      gh%cosa_u = c_loc(gridstruct%cosa_u); gh%sina_u = c_loc(gridstruct%sina_u); gh%rsin_u = c_loc(gridstruct%rsin_u)
      gh%divg_v = c_loc(gridstruct%divg_v); gh%del6_v = c_loc(gridstruct%del6_v)
      gh%rarea_c = c_loc(gridstruct%rarea_c); gh%fC = c_loc(gridstruct%fC)
      gh%cosa = c_loc(gridstruct%cosa);     gh%sina = c_loc(gridstruct%sina)
      gh%sin_sg = c_loc(gridstruct%sin_sg); gh%cos_sg = c_loc(gridstruct%cos_sg)
      fl%n_split = n_split;               fl%k_split = flagstruct%k_split;   fl%q_split = flagstruct%q_split
      fl%nord = flagstruct%nord;          fl%d4_bg = flagstruct%d4_bg;       fl%d2_bg = flagstruct%d2_bg
      fl%d2_bg_k1 = flagstruct%d2_bg_k1;  fl%d2_bg_k2 = flagstruct%d2_bg_k2; fl%dddmp = flagstruct%dddmp
      fl%vtdm4 = flagstruct%vtdm4;        fl%d_con = flagstruct%d_con;       fl%ke_bg = flagstruct%ke_bg
      fl%do_vort_damp = flagstruct%do_vort_damp; fl%use_logp = flagstruct%use_logp
      fl%use_old_omega = flagstruct%use_old_omega; fl%is_ideal_case = flagstruct%is_ideal_case
      fl%n_sponge = flagstruct%n_sponge
      fl%hord_mt = flagstruct%hord_mt;    fl%hord_vt = flagstruct%hord_vt;   fl%hord_tm = flagstruct%hord_tm
      fl%hord_dp = flagstruct%hord_dp;    fl%hord_tr = flagstruct%hord_tr
      fl%kord_tm = flagstruct%kord_tm;    fl%kord_mt = flagstruct%kord_mt;   fl%kord_wz = flagstruct%kord_wz
      fl%kord_tr = flagstruct%kord_tr;    fl%nord_tr = flagstruct%nord_tr;   fl%trdm2 = flagstruct%trdm2
      fl%a_imp = flagstruct%a_imp;        fl%p_fac = flagstruct%p_fac;       fl%ptop = ptop
      fl%m_split = max(1, flagstruct%m_split)
      fl%grav = grav;                     fl%akap = akap;                    fl%cp_air = cp
      ! rdgas: constants_mod's, as in the reference (fv3_flags carries it as its default)
      fl%adiabatic = flagstruct%adiabatic; fl%fill = flagstruct%fill
      fl%hydrostatic = hydrostatic;       fl%d_ext = flagstruct%d_ext;       fl%delt_max = flagstruct%delt_max
      fl%use_cond = thermostruct%use_cond; fl%moist_kappa = thermostruct%moist_kappa
      fl%beta = flagstruct%beta           ! du / dv live in the bound fv3_atmos between calls, like dyn_core's saved arrays (:278-283)
      fl%convert_ke = flagstruct%convert_ke
      fl%fast_tau_w_sec = flagstruct%fast_tau_w_sec; fl%RF_fast = flagstruct%RF_fast; fl%tau = flagstruct%tau   ! :536, :940, :1057-1060
      fl%rf_cutoff = flagstruct%rf_cutoff; fl%ks = ks
      fl%do_diss_est = flagstruct%do_diss_est
      fl%fill_dp = flagstruct%fill_dp                                      ! dyn_core.F90:820
      call fv3_host_init_grid(at, dom, gh, 0, fl, ak, bk)
      bound = .true.
    end subroutine
  end subroutine dyn_core

  !> fv_dynamics with the reference's argument list (model/fv_dynamics.F90:79-85) for an adiabatic-core call: T -> theta_v
  !> (:284-399), the k_split loop (dyn_core, tracer_2d, Lagrangian_to_Eulerian with last_step on the final cycle, :460-665),
  !> cubed_to_latlon (:911), over the resident fv3_fv_dynamics of fv3_host_mod.  Host arrays in, host arrays out, like dyn_core
  !> above.  Carried: consv_te (energy fixer), tau > 0 (Rayleigh_Super / Rayleigh_Friction), RF_fast, fast_tau_w_sec, use_cond /
  !> moist_kappa, consv_am, do_diss_est, fill_dp, beta > 0 and beta < -0.1; hybrid_z and ze0 are accepted and, as in the reference's
  !> Lagrangian_to_Eulerian (fv_mapz.F90:62, :128: declared, never read), without effect.  error stop: nesting / regional domains.
  subroutine fv_dynamics(npx, npy, npz, nq_tot, ng, bdt, consv_te, fill, &
                         reproduce_sum, kappa, cp_air, zvir, ptop, ks, ncnst, n_split, &
                         q_split, u0, v0, u, v, w, delz, hydrostatic, pt, delp, q, &
                         ps, pe, pk, peln, pkz, phis, q_con, omga, ua, va, uc, vc, &
                         ak, bk, mfx, mfy, cx, cy, ze0, hybrid_z, &
                         gridstruct, flagstruct, neststruct, thermostruct, idiag, bd, &
                         parent_grid, domain, inline_mp, heat_source, diss_est, time_total)
    real(c_double), intent(in) :: bdt, consv_te, kappa, cp_air, zvir, ptop
    real(c_double), intent(in), optional :: time_total
    integer, intent(in) :: npx, npy, npz, nq_tot, ng, ks, ncnst, n_split, q_split
    logical, intent(in) :: fill, reproduce_sum, hydrostatic, hybrid_z
    type(fv_grid_bounds_type), intent(in) :: bd
    real(c_double), intent(inout), dimension(bd%isd:, bd%jsd:, 1:) :: u0, v0
    real(c_double), intent(inout), target :: u(bd%isd:bd%ied, bd%jsd:bd%jed+1, npz), v(bd%isd:bd%ied+1, bd%jsd:bd%jed, npz)
    real(c_double), intent(inout), target :: w(bd%isd:, bd%jsd:, 1:)
    real(c_double), intent(inout), target :: pt(bd%isd:bd%ied, bd%jsd:bd%jed, npz), delp(bd%isd:bd%ied, bd%jsd:bd%jed, npz)
    real(c_double), intent(inout), target :: q(bd%isd:bd%ied, bd%jsd:bd%jed, npz, ncnst)
    real(c_double), intent(inout), target :: delz(bd%is:, bd%js:, 1:)
    real(c_double), intent(inout) :: ze0(bd%is:, bd%js:, 1:)
    real(c_double), intent(inout), target :: diss_est(bd%isd:bd%ied, bd%jsd:bd%jed, npz)
    real(c_double), intent(inout) :: heat_source(bd%isd:bd%ied, bd%jsd:bd%jed, npz)
    real(c_double), intent(inout), target :: ps(bd%isd:bd%ied, bd%jsd:bd%jed)
    real(c_double), intent(inout), target :: pe(bd%is-1:bd%ie+1, npz+1, bd%js-1:bd%je+1)
    real(c_double), intent(inout), target :: pk(bd%is:bd%ie, bd%js:bd%je, npz+1), peln(bd%is:bd%ie, npz+1, bd%js:bd%je)
    real(c_double), intent(inout), target :: pkz(bd%is:bd%ie, bd%js:bd%je, npz)
    real(c_double), intent(inout), target :: q_con(bd%isd:, bd%jsd:, 1:)
    real(c_double), intent(inout), target :: phis(bd%isd:bd%ied, bd%jsd:bd%jed), omga(bd%isd:bd%ied, bd%jsd:bd%jed, npz)
    real(c_double), intent(inout), target :: uc(bd%isd:bd%ied+1, bd%jsd:bd%jed, npz), vc(bd%isd:bd%ied, bd%jsd:bd%jed+1, npz)
    real(c_double), intent(inout), target, dimension(bd%isd:bd%ied, bd%jsd:bd%jed, npz) :: ua, va
    real(c_double), intent(in) :: ak(npz+1), bk(npz+1)
    type(inline_mp_type), intent(inout) :: inline_mp
    real(c_double), intent(inout), target :: mfx(bd%is:bd%ie+1, bd%js:bd%je, npz), mfy(bd%is:bd%ie, bd%js:bd%je+1, npz)
    real(c_double), intent(inout), target :: cx(bd%is:bd%ie+1, bd%jsd:bd%jed, npz), cy(bd%isd:bd%ied, bd%js:bd%je+1, npz)
    type(fv_grid_type), intent(inout), target :: gridstruct
    type(fv_flags_type), intent(inout) :: flagstruct
    type(fv_nest_type), intent(inout) :: neststruct
    type(domain2d), intent(inout) :: domain
    type(fv_atmos_type), pointer, intent(in) :: parent_grid
    type(fv_diag_type), intent(in) :: idiag
    type(fv_thermo_type), intent(inout) :: thermostruct

    real(c_double), allocatable, target :: w_c(:,:,:), delz_c(:,:,:), zs(:,:), qc_c(:,:,:)
    integer(c_size_t) :: nk, nk1
    integer :: nx, ny

    if (neststruct%nested .or. gridstruct%nested .or. gridstruct%regional .or. gridstruct%bounded_domain) &
      error stop 'fv_dynamics (fv3_dyn_core_mod): nested / regional domains are not built'
    if (gridstruct%grid_type < 3) then
      call fv_dynamics_sphere()
      return
    end if
    if ((thermostruct%use_cond .or. thermostruct%moist_kappa) .and. hydrostatic) &
      error stop 'fv_dynamics (fv3_dyn_core_mod): use_cond / moist_kappa are nonhydrostatic branches'
    if (thermostruct%use_cond .and. (size(q_con, 1) /= bd%ied - bd%isd + 1 .or. size(q_con, 3) < npz)) &
      error stop 'fv_dynamics (fv3_dyn_core_mod): use_cond needs q_con(isd:ied, jsd:jed, npz)'
    if (gridstruct%grid_type /= 4) error stop 'fv_dynamics (fv3_dyn_core_mod): grid_type = 3 is not built'
    ! hybrid_z: the reference hands it on to Lagrangian_to_Eulerian (fv_dynamics.F90:615), which declares it and never reads it
    ! (fv_mapz.F90:62, :128); ze0 likewise is not touched on this path -- both are accepted and, as there, without effect
    if (flagstruct%beta < 0.d0 .and. (hydrostatic .or. flagstruct%beta >= -0.1d0)) &
      error stop 'fv_dynamics (fv3_dyn_core_mod): beta < 0: one_grad_p (beta < -0.1) is built for the nonhydrostatic loop'
    if (flagstruct%consv_am .and. .not. (allocated(gridstruct%agrid) .and. allocated(gridstruct%l2c_u) .and. allocated(gridstruct%l2c_v) &
                                         .and. allocated(idiag%zxg))) &
      error stop 'fv_dynamics (fv3_dyn_core_mod): consv_am needs gridstruct%agrid, %l2c_u, %l2c_v and idiag%zxg'
    if (ng /= 3 .or. nq_tot > ncnst) error stop 'fv_dynamics (fv3_dyn_core_mod): ng = 3, nq_tot <= ncnst'
    if (.not. boundf) then
      call bind_context()
      if (fv3_domain_npes(domain) > 1) call fv3_host_comm_layout(atf, fv3_domain_pe(domain), fv3_domain_npes(domain), fv3_domain_layout_of(domain, 1), fv3_domain_layout_of(domain, 2), fv3_domain_comm_id(domain))
    end if
    if (atf%npz /= npz .or. atf%nq /= nq_tot .or. atf%ie /= bd%ie .or. atf%je /= bd%je) &
      error stop 'fv_dynamics (fv3_dyn_core_mod): the domain changed between calls'
    atf%fl%n_split = n_split; atf%fl%q_split = q_split
    nx = bd%ie - bd%is + 1; ny = bd%je - bd%js + 1
    nk = int(npz, c_size_t); nk1 = nk + 1

    call put(atf%u, c_loc(u), atf%nU*nk);        call put(atf%v, c_loc(v), atf%nV*nk)
    call put(atf%delp, c_loc(delp), atf%nA*nk);  call put(atf%pt, c_loc(pt), atf%nA*nk)
    call put(atf%phis, c_loc(phis), atf%nA)
    allocate(zs(bd%isd:bd%ied, bd%jsd:bd%jed))
    zs = phis * (1.d0 / atf%fl%grav)
    call put(atf%zs, c_loc(zs), atf%nA)
    if (.not. hydrostatic) then
      allocate(w_c(bd%isd:bd%ied, bd%jsd:bd%jed, npz), delz_c(bd%is:bd%ie, bd%js:bd%je, npz))
      w_c = w(bd%isd:bd%ied, bd%jsd:bd%jed, 1:npz); delz_c = delz(bd%is:bd%ie, bd%js:bd%je, 1:npz)
      call put(atf%w, c_loc(w_c), atf%nA*nk);    call put(atf%delz, c_loc(delz_c), atf%nCC*nk)
    end if
    if (nq_tot > 0) call put(atf%q, c_loc(q), atf%nA*nk*nq_tot)
    call put(atf%pkz, c_loc(pkz), atf%nCC*nk);   call put(atf%pk, c_loc(pk), atf%nCC*nk1)
    call put(atf%pe, c_loc(pe), int(nx+2, c_size_t)*nk1*(ny+2)); call put(atf%peln, c_loc(peln), atf%nCC*nk1)
    call put(atf%omga, c_loc(omga), atf%nA*nk)
    call fv3_check(fv3_sync(atf%ctx), 'fv3_sync')

    ! :284-399 T -> theta_v, :345 compute_total_energy, :362-375 Rayleigh_Friction, the k_split loop :460-665 with the energy fixer of
    ! its last remap, cubed_to_latlon :911
    atf%fl%adiabatic = flagstruct%adiabatic .or. zvir == 0.d0 .or. nq_tot == 0
    if (flagstruct%consv_am .and. .not. atf%consv_am) call bind_consv_am(atf)
    call fv3_fv_dynamics_call(atf, bdt, consv_te, merge(0.d0, flagstruct%tau, flagstruct%RF_fast), flagstruct%rf_cutoff, zvir, flagstruct%c2l_ord, flagstruct%moist_phys, &
                              6.3712d6)

    call fv3_check(fv3_sync(atf%ctx), 'fv3_sync')
    call get(c_loc(u), atf%u, atf%nU*nk);        call get(c_loc(v), atf%v, atf%nV*nk)
    call get(c_loc(delp), atf%delp, atf%nA*nk);  call get(c_loc(pt), atf%pt, atf%nA*nk)
    if (.not. hydrostatic) then
      call get(c_loc(w_c), atf%w, atf%nA*nk);    call get(c_loc(delz_c), atf%delz, atf%nCC*nk)
    end if
    if (nq_tot > 0) call get(c_loc(q), atf%q, atf%nA*nk*nq_tot)
    call get(c_loc(ps), atf%ps, atf%nA)
    call get(c_loc(pkz), atf%pkz, atf%nCC*nk);   call get(c_loc(pk), atf%pk, atf%nCC*nk1)
    call get(c_loc(pe), atf%pe, int(nx+2, c_size_t)*nk1*(ny+2)); call get(c_loc(peln), atf%peln, atf%nCC*nk1)
    call get(c_loc(omga), atf%omga, atf%nA*nk);  call get(c_loc(ua), atf%ua, atf%nA*nk); call get(c_loc(va), atf%va, atf%nA*nk)
    call get(c_loc(uc), atf%uc, atf%nV*nk);      call get(c_loc(vc), atf%vc, atf%nU*nk)
    call get(c_loc(mfx), atf%mfx, atf%nFX*nk);   call get(c_loc(mfy), atf%mfy, atf%nFY*nk)
    call get(c_loc(cx), atf%cx, atf%nCX*nk);     call get(c_loc(cy), atf%cy, atf%nCY*nk)
    if (flagstruct%do_diss_est) call get(c_loc(diss_est), atf%diss_est, atf%nA*nk)   ! zeroed at the first cycle, summed over all of them
    if (thermostruct%use_cond) then            ! q_con as moist_cv left it (fv_dynamics.F90:305-317 and the remaps)
      allocate(qc_c(bd%isd:bd%ied, bd%jsd:bd%jed, npz))
      call get(c_loc(qc_c), atf%q_con, atf%nA*nk)
    end if
    call fv3_check(fv3_sync(atf%ctx), 'fv3_sync')
    if (.not. hydrostatic) then
      w(bd%isd:bd%ied, bd%jsd:bd%jed, 1:npz) = w_c; delz(bd%is:bd%ie, bd%js:bd%je, 1:npz) = delz_c
    end if
    if (thermostruct%use_cond) q_con(bd%isd:bd%ied, bd%jsd:bd%jed, 1:npz) = qc_c

  contains

    subroutine put(d, h, n)
      type(c_ptr), intent(in) :: d, h
      integer(c_size_t), intent(in) :: n
      call fv3_check(fv3_memcpy_h2d(atf%ctx, d, h, n * 8_c_size_t), 'fv3_memcpy_h2d')
    end subroutine

    subroutine get(h, d, n)
      type(c_ptr), intent(in) :: d, h
      integer(c_size_t), intent(in) :: n
      call fv3_check(fv3_memcpy_d2h(atf%ctx, h, d, n * 8_c_size_t), 'fv3_memcpy_d2h')
    end subroutine

    !> flagstruct%consv_am: cos(agrid(:,:,2)), l2c_u / l2c_v (padded with zeros to the halo'd U / V shapes the kernels index) and
    !> idiag%zxg go to the resident loop once (fv_dynamics.F90:358-361, :747-800, :1266-1314)
    subroutine bind_consv_am(a)
      type(fv3_atmos), intent(inout) :: a
      real(c_double), allocatable, target :: cl(:,:), lu(:,:), lv(:,:)
      allocate(cl(bd%isd:bd%ied, bd%jsd:bd%jed), lu(bd%isd:bd%ied, bd%jsd:bd%jed+1), lv(bd%isd:bd%ied+1, bd%jsd:bd%jed))
      cl = cos(gridstruct%agrid(bd%isd:bd%ied, bd%jsd:bd%jed, 2))
      lu = 0.d0; lv = 0.d0
      lu(bd%is:bd%ie, bd%js:bd%je+1) = gridstruct%l2c_u
      lv(bd%is:bd%ie+1, bd%js:bd%je) = gridstruct%l2c_v
      if (.not. allocated(gridstruct%agrid) .or. .not. allocated(gridstruct%l2c_u) .or. .not. allocated(gridstruct%l2c_v) .or. &
          .not. allocated(idiag%zxg)) error stop 'fv_dynamics (fv3_dyn_core_mod): consv_am reads gridstruct%agrid, %l2c_u, %l2c_v and idiag%zxg'
      call fv3_host_set_consv_am(a, cl, lu, lv, idiag%zxg)
    end subroutine

    !> grid_type < 3: a tile of the cubed sphere.  The reference calls fv_dynamics once per tile a PE holds -- with one tile per PE
    !> that is one call per PE, all PEs at the same time, and the halo updates inside meet in mpp_update_domains.  Here the tiles one
    !> process holds (domain%face_rank) are given one call after the other, and the exchanges need all of them: every call binds
    !> (first time) and uploads its tile; the call of the LAST tile this process holds runs the step for all of them -- compute_total_
    !> energy, theta_v, Rayleigh_Super, the k_split loop with the cube-edge exchange behind fv3_cube_halo_* (RCCL between processes),
    !> the energy fixer, cubed_to_latlon: fv3_sphere_fv_dynamics_call -- and writes the results into the arrays of every tile's call.
    !> Those arrays are the model's state (Atm(n)%u ...): contiguous, with the reference's extents, alive until that last call.
    subroutine fv_dynamics_sphere()
      type(fv3_flags) :: fl
      integer :: slot, nloc, sl
      if (ng /= 3 .or. nq_tot > ncnst) error stop 'fv_dynamics (fv3_dyn_core_mod): ng = 3, nq_tot <= ncnst'
      if (fv3_domain_tile(domain) < 1 .or. fv3_domain_tile(domain) > 6) error stop 'fv_dynamics (fv3_dyn_core_mod): fv3_domain_tile(domain) must be 1 .. 6'
      if (bd%is /= 1 .or. bd%js /= 1 .or. bd%ie /= npx - 1 .or. bd%je /= npy - 1) &
        error stop 'fv_dynamics (fv3_dyn_core_mod): one whole tile per context (layout 1 x 1 per tile)'
      nloc = fv3_tiles_held(domain, 6)
      slot = fv3_tiles_held(domain, fv3_domain_tile(domain))
      if (fv3_domain_tile_pe(domain, fv3_domain_tile(domain)) /= fv3_domain_pe(domain)) error stop 'fv_dynamics (fv3_dyn_core_mod): this PE does not hold fv3_domain_tile(domain)'
      nx = bd%ie - bd%is + 1; ny = bd%je - bd%js + 1
      nk = int(npz, c_size_t); nk1 = nk + 1
      if (.not. hydrostatic) then
        if (.not. (is_contiguous(w) .and. is_contiguous(delz)) .or. size(w, 1) /= nx + 6 .or. size(w, 2) /= ny + 6 .or. &
            size(w, 3) /= npz .or. size(delz, 1) /= nx .or. size(delz, 2) /= ny .or. size(delz, 3) /= npz) &
          error stop 'fv_dynamics (fv3_dyn_core_mod): w / delz must be the whole contiguous arrays of the tile'
      end if
      if (.not. bound_s(slot)) then
        call flags_of(flagstruct, fl)
        fl%ks = ks
        fl%n_split = n_split; fl%q_split = q_split; fl%ptop = ptop; fl%akap = kappa; fl%cp_air = cp_air
        fl%hydrostatic = hydrostatic; fl%fill = fill; fl%r_vir = zvir
        call moist_flags_of(flagstruct, thermostruct, fl)
        fl%do_diss_est = flagstruct%do_diss_est
        call bind_sphere_tile(sps, slot, fv3_domain_tile(domain), npx, npy, npz, nq_tot, bd, gridstruct, flagstruct, fl, ak, bk)
        bound_s(slot) = .true.
      end if
      if (flagstruct%consv_am .and. .not. sps%f(slot)%consv_am) call bind_consv_am(sps%f(slot))   ! this tile's coslat, l2c_u, l2c_v, zxg
      associate (at => sps%f(slot))
        if (at%npz /= npz .or. at%nq /= nq_tot .or. at%ie /= bd%ie .or. at%je /= bd%je) &
          error stop 'fv_dynamics (fv3_dyn_core_mod): the domain changed between calls'
        at%fl%n_split = n_split; at%fl%q_split = q_split
        call puts(at, at%u, c_loc(u), at%nU*nk);        call puts(at, at%v, c_loc(v), at%nV*nk)
        call puts(at, at%delp, c_loc(delp), at%nA*nk);  call puts(at, at%pt, c_loc(pt), at%nA*nk)
        call puts(at, at%phis, c_loc(phis), at%nA)
        allocate(zs(bd%isd:bd%ied, bd%jsd:bd%jed))
        zs = phis * (1.d0 / at%fl%grav)
        call puts(at, at%zs, c_loc(zs), at%nA)
        if (.not. hydrostatic) then
          call puts(at, at%w, c_loc(w), at%nA*nk);      call puts(at, at%delz, c_loc(delz), at%nCC*nk)
        end if
        if (nq_tot > 0) call puts(at, at%q, c_loc(q), at%nA*nk*nq_tot)
        call puts(at, at%pkz, c_loc(pkz), at%nCC*nk);   call puts(at, at%pk, c_loc(pk), at%nCC*nk1)
        call puts(at, at%pe, c_loc(pe), int(nx+2, c_size_t)*nk1*(ny+2)); call puts(at, at%peln, c_loc(peln), at%nCC*nk1)
        call puts(at, at%omga, c_loc(omga), at%nA*nk)
        call fv3_check(fv3_sync(at%ctx), 'fv3_sync')
      end associate
      tps(slot)%u = c_loc(u); tps(slot)%v = c_loc(v); tps(slot)%pt = c_loc(pt); tps(slot)%delp = c_loc(delp)
      tps(slot)%w = c_null_ptr; tps(slot)%delz = c_null_ptr
      if (.not. hydrostatic) then
        tps(slot)%w = c_loc(w); tps(slot)%delz = c_loc(delz)
      end if
      tps(slot)%q = c_loc(q); tps(slot)%ps = c_loc(ps); tps(slot)%pe = c_loc(pe); tps(slot)%pk = c_loc(pk)
      tps(slot)%peln = c_loc(peln); tps(slot)%pkz = c_loc(pkz); tps(slot)%omga = c_loc(omga)
      tps(slot)%ua = c_loc(ua); tps(slot)%va = c_loc(va); tps(slot)%uc = c_loc(uc); tps(slot)%vc = c_loc(vc)
      tps(slot)%mfx = c_loc(mfx); tps(slot)%mfy = c_loc(mfy); tps(slot)%cx = c_loc(cx); tps(slot)%cy = c_loc(cy)
      tps(slot)%q_con = c_null_ptr
      if (thermostruct%use_cond) then     ! q_con as moist_cv leaves it (fv_dynamics.F90:305-317 and the remaps) goes back to the model's array
        if (.not. is_contiguous(q_con) .or. size(q_con, 1) /= nx + 6 .or. size(q_con, 2) /= ny + 6 .or. size(q_con, 3) /= npz) &
          error stop 'fv_dynamics (fv3_dyn_core_mod): use_cond on the cubed sphere needs q_con(isd:ied, jsd:jed, npz), contiguous'
        tps(slot)%q_con = c_loc(q_con)
      end if
      tps(slot)%diss_est = c_loc(diss_est)           ! do_diss_est: zeroed at the first cycle of the call, summed over all of them
      if (slot < nloc) return                        ! the step runs in the call of the last tile this process holds

      if (.not. comm_s) then
        if (fv3_domain_npes(domain) > 1) then
          call fv3_sphere_comm(sps, fv3_domain_pe(domain), fv3_domain_npes(domain), fv3_domain_tile_pelist(domain), fv3_domain_comm_id(domain))
        else
          call fv3_sphere_comm(sps, 0, 1, fv3_domain_tile_pelist(domain))
        end if
        comm_s = .true.
      end if
      call fv3_sphere_fv_dynamics_call(sps, bdt, fv3_domain_npes(domain), consv_te, merge(0.d0, flagstruct%tau, flagstruct%RF_fast), flagstruct%rf_cutoff, zvir, &
                                       flagstruct%c2l_ord, flagstruct%moist_phys, 6.3712d6)         ! constants_mod: radius
      do sl = 1, nloc
        associate (at => sps%f(sl), tp => tps(sl))
          call fv3_check(fv3_sync(at%ctx), 'fv3_sync')
          call gets(at, tp%u, at%u, at%nU*nk);        call gets(at, tp%v, at%v, at%nV*nk)
          call gets(at, tp%delp, at%delp, at%nA*nk);  call gets(at, tp%pt, at%pt, at%nA*nk)
          if (.not. hydrostatic) then
            call gets(at, tp%w, at%w, at%nA*nk);      call gets(at, tp%delz, at%delz, at%nCC*nk)
          end if
          if (nq_tot > 0) call gets(at, tp%q, at%q, at%nA*nk*nq_tot)
          call gets(at, tp%ps, at%ps, at%nA)
          call gets(at, tp%pkz, at%pkz, at%nCC*nk);   call gets(at, tp%pk, at%pk, at%nCC*nk1)
          call gets(at, tp%pe, at%pe, int(nx+2, c_size_t)*nk1*(ny+2)); call gets(at, tp%peln, at%peln, at%nCC*nk1)
          call gets(at, tp%omga, at%omga, at%nA*nk);  call gets(at, tp%ua, at%ua, at%nA*nk); call gets(at, tp%va, at%va, at%nA*nk)
          call gets(at, tp%uc, at%uc, at%nV*nk);      call gets(at, tp%vc, at%vc, at%nU*nk)
          call gets(at, tp%mfx, at%mfx, at%nFX*nk);   call gets(at, tp%mfy, at%mfy, at%nFY*nk)
          call gets(at, tp%cx, at%cx, at%nCX*nk);     call gets(at, tp%cy, at%cy, at%nCY*nk)
          if (c_associated(tp%q_con)) call gets(at, tp%q_con, at%q_con, at%nA*nk)
          if (flagstruct%do_diss_est) call gets(at, tp%diss_est, at%diss_est, at%nA*nk)
          call fv3_check(fv3_sync(at%ctx), 'fv3_sync')
        end associate
      end do
    end subroutine

    subroutine puts(a, d, h, n)
      type(fv3_atmos), intent(in) :: a
      type(c_ptr), intent(in) :: d, h
      integer(c_size_t), intent(in) :: n
      call fv3_check(fv3_memcpy_h2d(a%ctx, d, h, n * 8_c_size_t), 'fv3_memcpy_h2d')
    end subroutine

    subroutine gets(a, h, d, n)
      type(fv3_atmos), intent(in) :: a
      type(c_ptr), intent(in) :: d, h
      integer(c_size_t), intent(in) :: n
      call fv3_check(fv3_memcpy_d2h(a%ctx, h, d, n * 8_c_size_t), 'fv3_memcpy_d2h')
    end subroutine

    subroutine bind_context()
      type(fv3_domain) :: dom
      type(fv3_grid_host) :: gh
      type(fv3_flags) :: fl
      dom%is = bd%is; dom%ie = bd%ie; dom%js = bd%js; dom%je = bd%je; dom%ng = 3
      dom%npx = npx; dom%npy = npy; dom%npz = npz; dom%grid_type = gridstruct%grid_type
      dom%do_diss_est = merge(1, 0, flagstruct%do_diss_est); dom%prevent_diss_cooling = merge(1, 0, flagstruct%prevent_diss_cooling)
      dom%stretched_grid = merge(1, 0, gridstruct%stretched_grid); dom%lim_fac = flagstruct%lim_fac
      call grid_host_of(gridstruct, gh)
      call flags_of(flagstruct, fl)
      fl%do_diss_est = flagstruct%do_diss_est
      fl%ks = ks
      fl%n_split = n_split; fl%q_split = q_split; fl%ptop = ptop; fl%akap = kappa; fl%cp_air = cp_air
      fl%hydrostatic = hydrostatic; fl%fill = fill; fl%r_vir = zvir
      call moist_flags_of(flagstruct, thermostruct, fl)
      call fv3_host_init_grid(atf, dom, gh, nq_tot, fl, ak, bk)
      boundf = .true.
    end subroutine
  end subroutine fv_dynamics

  subroutine fv_dynamics_end()
    if (boundf) call fv3_host_final(atf)
    boundf = .false.
    if (any(bound_s)) call fv3_sphere_final(sps)
    bound_s = .false.; comm_s = .false.
  end subroutine

  !> one tile of the cubed sphere: context (grid_type < 3), gridstruct with the cubed members in the library's layouts (a11 .. a22 on
  !> the A layout, the unit vectors with the component last; fv3_grid_upload_cubed copies them), device arrays
  subroutine bind_sphere_tile(sp, slot, tile, npx, npy, npz, nq, bd, gridstruct, flagstruct, fl, ak, bk)
    type(fv3_sphere), intent(inout) :: sp
    integer, intent(in) :: slot, tile, npx, npy, npz, nq
    type(fv_grid_bounds_type), intent(in) :: bd
    type(fv_grid_type), intent(inout), target :: gridstruct
    type(fv_flags_type), intent(in) :: flagstruct
    type(fv3_flags), intent(in) :: fl
    real(c_double), intent(in) :: ak(npz+1), bk(npz+1)
    type(fv3_domain) :: dom
    type(fv3_grid_host) :: gh
    type(fv3_grid_cubed) :: gc
    real(c_double), allocatable, target :: a4(:,:,:), ecp(:,:,:,:), en1p(:,:,:), en2p(:,:,:)
    integer :: t
    dom%is = bd%is; dom%ie = bd%ie; dom%js = bd%js; dom%je = bd%je; dom%ng = 3
    dom%npx = npx; dom%npy = npy; dom%npz = npz; dom%grid_type = gridstruct%grid_type
    dom%do_diss_est = merge(1, 0, flagstruct%do_diss_est); dom%prevent_diss_cooling = merge(1, 0, flagstruct%prevent_diss_cooling)
    dom%stretched_grid = merge(1, 0, gridstruct%stretched_grid); dom%lim_fac = flagstruct%lim_fac
    call grid_host_of(gridstruct, gh)
    allocate(a4(bd%isd:bd%ied, bd%jsd:bd%jed, 4), ecp(bd%isd:bd%ied, bd%jsd:bd%jed, 3, 2))
    allocate(en1p(bd%is:bd%ie, bd%js:bd%je+1, 3), en2p(bd%is:bd%ie+1, bd%js:bd%je, 3))
    a4 = 0.d0
    a4(bd%is-1:bd%ie+1, bd%js-1:bd%je+1, 1) = gridstruct%a11; a4(bd%is-1:bd%ie+1, bd%js-1:bd%je+1, 2) = gridstruct%a12
    a4(bd%is-1:bd%ie+1, bd%js-1:bd%je+1, 3) = gridstruct%a21; a4(bd%is-1:bd%ie+1, bd%js-1:bd%je+1, 4) = gridstruct%a22
    do t = 1, 3
      ecp(:, :, t, 1) = gridstruct%ec1(t, :, :); ecp(:, :, t, 2) = gridstruct%ec2(t, :, :)
      en1p(:, :, t) = gridstruct%en1(t, :, :);   en2p(:, :, t) = gridstruct%en2(t, :, :)
    end do
    gc%edge_w = c_loc(gridstruct%edge_w); gc%edge_e = c_loc(gridstruct%edge_e)
    gc%edge_s = c_loc(gridstruct%edge_s); gc%edge_n = c_loc(gridstruct%edge_n)
    gc%rsina = c_loc(gridstruct%rsina)
    if (gridstruct%corner_f(1) < 0.d0) call corner_factors(gridstruct, bd, npx, npy)
    gc%corner_f = gridstruct%corner_f
    gc%a11 = c_loc(a4(bd%isd, bd%jsd, 1)); gc%a12 = c_loc(a4(bd%isd, bd%jsd, 2))
    gc%a21 = c_loc(a4(bd%isd, bd%jsd, 3)); gc%a22 = c_loc(a4(bd%isd, bd%jsd, 4))
    gc%ec1 = c_loc(ecp(bd%isd, bd%jsd, 1, 1)); gc%ec2 = c_loc(ecp(bd%isd, bd%jsd, 1, 2))
    gc%en1 = c_loc(en1p(bd%is, bd%js, 1));     gc%en2 = c_loc(en2p(bd%is, bd%js, 1))
    call fv3_sphere_init_face(sp, slot, tile - 1, dom, gh, gc, nq, fl, ak, bk)
  end subroutine

  !> the extrap_corner factors x1 / (x2 - x1) of a2b_ord4 (model/a2b_edge.F90:83-112 for the four corners, extrap_corner :452-462)
  !> from grid / agrid, with great_circle_dist of fv_grid_utils.F90:2568-2591 (radius 1): corners sw, se, ne, nw, the three
  !> (inner, outer) cell-centre pairs in the reference's order
  subroutine corner_factors(gridstruct, bd, npx, npy)
    type(fv_grid_type), intent(inout) :: gridstruct
    type(fv_grid_bounds_type), intent(in) :: bd
    integer, intent(in) :: npx, npy
    integer :: n
    n = npx
    if (npx /= npy) error stop 'corner_factors: npx = npy'
    gridstruct%corner_f(1)  = fac(1, 1, 1, 1, 2, 2);         gridstruct%corner_f(2)  = fac(1, 1, 0, 1, -1, 2)
    gridstruct%corner_f(3)  = fac(1, 1, 1, 0, 2, -1)
    gridstruct%corner_f(4)  = fac(n, 1, n-1, 1, n-2, 2);     gridstruct%corner_f(5)  = fac(n, 1, n-1, 0, n-2, -1)
    gridstruct%corner_f(6)  = fac(n, 1, n, 1, n+1, 2)
    gridstruct%corner_f(7)  = fac(n, n, n-1, n-1, n-2, n-2); gridstruct%corner_f(8)  = fac(n, n, n, n-1, n+1, n-2)
    gridstruct%corner_f(9)  = fac(n, n, n-1, n, n-2, n+1)
    gridstruct%corner_f(10) = fac(1, n, 1, n-1, 2, n-2);     gridstruct%corner_f(11) = fac(1, n, 0, n-1, -1, n-2)
    gridstruct%corner_f(12) = fac(1, n, 1, n, 2, n+1)
  contains
    real(c_double) function fac(i0, j0, ia, ja, ib, jb)
      integer, intent(in) :: i0, j0, ia, ja, ib, jb
      real(c_double) :: x1, x2
      x1 = gcd(gridstruct%agrid(ia, ja, :), gridstruct%grid(i0, j0, :), .true.)      ! (the members carry the reference's bounds)
      x2 = gcd(gridstruct%agrid(ib, jb, :), gridstruct%grid(i0, j0, :), .true.)
      fac = x1 / (x2 - x1)
    end function
    real(c_double) function gcd(q1, q2, dummy)
      real(c_double), intent(in) :: q1(2), q2(2)
      logical, intent(in) :: dummy
      real(c_double) :: p1, p2, dp, dl
      p1 = q1(2); p2 = q2(2)
      dp = sin(0.5d0 * (p1 - p2)); dl = sin(0.5d0 * (q1(1) - q2(1)))
      gcd = 2.d0 * asin(sqrt(dp * dp + cos(p1) * cos(p2) * dl * dl))
    end function
  end subroutine

  !> gridstruct members by address -> the host-pointer structure of fv3_grid_upload
  subroutine grid_host_of(gridstruct, gh)
    type(fv_grid_type), intent(in), target :: gridstruct
    type(fv3_grid_host), intent(out) :: gh
    gh%da_min = gridstruct%da_min;        gh%da_min_c = gridstruct%da_min_c
    gh%area = c_loc(gridstruct%area);     gh%rarea = c_loc(gridstruct%rarea)
    gh%dxa = c_loc(gridstruct%dxa);       gh%dya = c_loc(gridstruct%dya)
    gh%rdxa = c_loc(gridstruct%rdxa);     gh%rdya = c_loc(gridstruct%rdya)
    gh%cosa_s = c_loc(gridstruct%cosa_s); gh%rsin2 = c_loc(gridstruct%rsin2);   gh%f0 = c_loc(gridstruct%f0)
    gh%dx = c_loc(gridstruct%dx);         gh%rdx = c_loc(gridstruct%rdx)
    gh%dyc = c_loc(gridstruct%dyc);       gh%rdyc = c_loc(gridstruct%rdyc)
    gh%cosa_v = c_loc(gridstruct%cosa_v); gh%sina_v = c_loc(gridstruct%sina_v); gh%rsin_v = c_loc(gridstruct%rsin_v)
    gh%divg_u = c_loc(gridstruct%divg_u); gh%del6_u = c_loc(gridstruct%del6_u)
    gh%dy = c_loc(gridstruct%dy);         gh%rdy = c_loc(gridstruct%rdy)
    gh%dxc = c_loc(gridstruct%dxc);       gh%rdxc = c_loc(gridstruct%rdxc)
    gh%cosa_u = c_loc(gridstruct%cosa_u); gh%sina_u = c_loc(gridstruct%sina_u); gh%rsin_u = c_loc(gridstruct%rsin_u)
    gh%divg_v = c_loc(gridstruct%divg_v); gh%del6_v = c_loc(gridstruct%del6_v)
    gh%rarea_c = c_loc(gridstruct%rarea_c); gh%fC = c_loc(gridstruct%fC)
    gh%cosa = c_loc(gridstruct%cosa);     gh%sina = c_loc(gridstruct%sina)
    gh%sin_sg = c_loc(gridstruct%sin_sg); gh%cos_sg = c_loc(gridstruct%cos_sg)
  end subroutine

  !> thermostruct%use_cond / moist_kappa (the reference's defaults, fv_arrays.F90:1226-1227): the water species as fv_dynamics.F90:275-283
  !> finds them -- get_tracer_index -- and the heat capacities of fv_thermodynamics' moist_cv (cv_vap = 3 rvgas, c_liq, c_ice)
  subroutine moist_flags_of(flagstruct, thermostruct, fl)
    type(fv_flags_type), intent(in) :: flagstruct
    type(fv_thermo_type), intent(in) :: thermostruct
    type(fv3_flags), intent(inout) :: fl
    fl%use_cond = thermostruct%use_cond; fl%moist_kappa = thermostruct%moist_kappa
    if (fl%use_cond .or. fl%moist_kappa) then
      fl%moist%nwat = int(flagstruct%nwat, c_int)
      fl%moist%sphum = int(max(0, get_tracer_index(MODEL_ATMOS, 'sphum')), c_int)
      fl%moist%liq_wat = int(max(0, get_tracer_index(MODEL_ATMOS, 'liq_wat')), c_int)
      fl%moist%ice_wat = int(max(0, get_tracer_index(MODEL_ATMOS, 'ice_wat')), c_int)
      fl%moist%rainwat = int(max(0, get_tracer_index(MODEL_ATMOS, 'rainwat')), c_int)
      fl%moist%snowwat = int(max(0, get_tracer_index(MODEL_ATMOS, 'snowwat')), c_int)
      fl%moist%graupel = int(max(0, get_tracer_index(MODEL_ATMOS, 'graupel')), c_int)
      if (fl%moist%sphum < 1) error stop 'fv_dynamics (fv3_dyn_core_mod): use_cond / moist_kappa need the index of sphum (fv3_register_tracer_index)'
      fl%moist%cv_vap = 3.d0 * 461.50d0; fl%moist%c_liq = 4.218d3; fl%moist%c_ice = 2.106d3
    end if
  end subroutine

  !> flagstruct -> the host's fv3_flags (the members with the same names)
  subroutine flags_of(flagstruct, fl)
    type(fv_flags_type), intent(in) :: flagstruct
    type(fv3_flags), intent(inout) :: fl
    fl%k_split = flagstruct%k_split;   fl%q_split = flagstruct%q_split
    fl%nord = flagstruct%nord;          fl%d4_bg = flagstruct%d4_bg;       fl%d2_bg = flagstruct%d2_bg
    fl%d2_bg_k1 = flagstruct%d2_bg_k1;  fl%d2_bg_k2 = flagstruct%d2_bg_k2; fl%dddmp = flagstruct%dddmp
    fl%vtdm4 = flagstruct%vtdm4;        fl%d_con = flagstruct%d_con;       fl%ke_bg = flagstruct%ke_bg
    fl%do_vort_damp = flagstruct%do_vort_damp; fl%use_logp = flagstruct%use_logp
    fl%use_old_omega = flagstruct%use_old_omega; fl%is_ideal_case = flagstruct%is_ideal_case
    fl%n_sponge = flagstruct%n_sponge
    fl%hord_mt = flagstruct%hord_mt;    fl%hord_vt = flagstruct%hord_vt;   fl%hord_tm = flagstruct%hord_tm
    fl%hord_dp = flagstruct%hord_dp;    fl%hord_tr = flagstruct%hord_tr
    fl%kord_tm = flagstruct%kord_tm;    fl%kord_mt = flagstruct%kord_mt;   fl%kord_wz = flagstruct%kord_wz
    fl%kord_tr = flagstruct%kord_tr;    fl%nord_tr = flagstruct%nord_tr;   fl%trdm2 = flagstruct%trdm2
    fl%a_imp = flagstruct%a_imp;        fl%p_fac = flagstruct%p_fac;       fl%m_split = max(1, flagstruct%m_split)
    fl%adiabatic = flagstruct%adiabatic; fl%fill = flagstruct%fill
    fl%d_ext = flagstruct%d_ext;        fl%delt_max = flagstruct%delt_max
    fl%beta = flagstruct%beta
    fl%convert_ke = flagstruct%convert_ke
    fl%fast_tau_w_sec = flagstruct%fast_tau_w_sec; fl%RF_fast = flagstruct%RF_fast; fl%tau = flagstruct%tau
    fl%rf_cutoff = flagstruct%rf_cutoff
    fl%fill_dp = flagstruct%fill_dp                                        ! dyn_core.F90:820
  end subroutine

  !> lazy = .true.: dyn_core (doubly periodic domain) copies a caller's array to the device only when the caller declared a write to it
  !> (fv3_host_touched) or it has never been seen, and leaves the results on the device until the caller asks (fv3_host_fetch)
  subroutine fv3_dyn_core_registry(lazy)
    logical, intent(in) :: lazy
    registry_lazy = lazy
  end subroutine

  !> the caller wrote this host array since the last dyn_core call (c_loc of the array; c_null_ptr: every array)
  subroutine fv3_host_touched(host)
    type(c_ptr), intent(in) :: host
    if (bound) call fv3_check(fv3_registry_host_touched(at%ctx, host), 'fv3_registry_host_touched')
  end subroutine

  !> bring this host array up to date before the caller reads it (c_null_ptr: every array whose host copy is stale)
  subroutine fv3_host_fetch(host)
    type(c_ptr), intent(in) :: host
    if (bound) call fv3_check(fv3_registry_fetch(at%ctx, host), 'fv3_registry_fetch')
  end subroutine

  subroutine fv3_dyn_core_registry_stats(out4)
    integer(c_long_long), intent(out) :: out4(4)   ! h2d copies, h2d skipped, d2h copies, d2h deferred
    out4 = 0
    if (bound) call fv3_check(fv3_registry_stats(at%ctx, out4), 'fv3_registry_stats')
  end subroutine

  !> release the context dyn_core bound at its first call
  subroutine dyn_core_end()
    if (bound) call fv3_host_final(at)
    bound = .false.
    if (any(bound_d)) call fv3_sphere_final(spd)
    bound_d = .false.; comm_d = .false.
  end subroutine

end module fv3_dyn_core_mod
