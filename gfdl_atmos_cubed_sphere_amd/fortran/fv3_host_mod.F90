!> Fortran host orchestration over the C ABI of libfv3_mi355x.so (include/fv3_mi355x.h): the counterparts of
!> dyn_core (model/dyn_core.F90:94-1393, nonhydrostatic and hydrostatic branches, d_con heating, non-nested, grid_type = 4), of tracer_2d's host part
!> (model/fv_tracer2d.F90:382-417, :466-541) and of the k_split loop of fv_dynamics (model/fv_dynamics.F90:460-665),
!> written the way a maintainer would rewrite those routines around the kernels: the same order of calls and halo
!> updates (cited line by line), every field device resident (type(c_ptr) handles from fv3_malloc), the fields that
!> d_sw / update_dz_d / tracer_2d update in place in the reference held as ping-pong pairs.
!>
!> One rank of a doubly periodic domain: the group halo updates are fv3_halo_fill_periodic, or (FV3_HOST_COMM=1) the
!> exchange behind the C ABI, fv3_halo_start / fv3_halo_complete over the context's RCCL communicator -- the form a
!> several-rank run uses with the neighbour ranks in to / from (see INTEGRATION.md).
!>
!> Not reproduced (off in every BASELINE config): nesting / regional BCs, breed_vortex_inline, do_fast_phys, Ray_fast
!> (beta < -0.1 -- one_grad_p in the nonhydrostatic loop -- is built).  The argument lists are this module's own (type fv3_atmos holds the device handles the
!> reference keeps in fv_atmos_type / dyn_core's work arrays); INTEGRATION.md maps them to the reference's call sites,
!> and fv3_dyn_core_mod.F90 puts dyn_core's own argument list (host arrays, gridstruct, flagstruct, bd) in front of
!> fv3_dyn_core for callers that keep their state on the host.
module fv3_host_mod
  use iso_c_binding
  use fv3_mi355x_mod
  implicit none
  private
  public :: fv3_flags, fv3_atmos
  public :: fv3_host_halo, fv3_host_init, fv3_host_init_grid, fv3_host_final, fv3_host_upload, fv3_host_download, fv3_host_comm_layout
  public :: fv3_host_set_consv_am
  public :: fv3_dyn_core, fv3_dyn_core_hydrostatic, fv3_tracer_2d, fv3_fv_dynamics, fv3_fv_dynamics_call
  public :: dmalloc, dzero, swap, upload_levels, host_n_con, KIND_A, KIND_U, KIND_V, KIND_B    ! shared with fv3_sphere_mod
  public :: inline_q_begin, inline_q_end, host_fast_tau_w, host_ray_fast, set_condensate, diss_est_begin

  integer(c_int), parameter :: KIND_A = 0, KIND_U = 1, KIND_V = 2, KIND_B = 3
  integer, parameter :: NG = 3

  !> the fv_flags_type members the substep reads (defaults: model/fv_arrays.F90:207-906)
  type fv3_flags
    integer :: n_split = 1, k_split = 1, q_split = 0
    integer :: nord = 1
    real(c_double) :: d4_bg = 0.16d0, d2_bg = 0.d0, d2_bg_k1 = 0.20d0, d2_bg_k2 = 0.015d0
    real(c_double) :: dddmp = 0.d0, vtdm4 = 0.d0, d_con = 0.d0, ke_bg = 0.d0
    logical :: do_vort_damp = .false., use_logp = .false., use_old_omega = .true., is_ideal_case = .false.
    integer :: n_sponge = 1
    integer :: hord_mt = 10, hord_vt = 10, hord_tm = 10, hord_dp = 10, hord_tr = 8
    integer :: kord_tm = -8, kord_mt = 8, kord_wz = 8, kord_tr = 8
    integer :: nord_tr = 0
    real(c_double) :: trdm2 = 0.d0
    real(c_double) :: a_imp = 1.d0, p_fac = 0.05d0
    integer :: m_split = 1                            ! :548; the sub-steps of RIM_2D (a_imp <= 0.5)
    real(c_double) :: ptop = 300.d0
    real(c_double) :: grav = 9.80d0, rdgas = 287.04d0, akap = 2.d0/7.d0, cp_air = 287.04d0/(2.d0/7.d0)   ! constants_mod
    real(c_double) :: r_vir = 0.6077d0, t_min = 184.d0
    logical :: adiabatic = .true., fill = .false.
    logical :: hydrostatic = .false.                  ! fv_arrays.F90:366
    real(c_double) :: d_ext = 0.02d0, delt_max = 1.d0 ! :452, :441
    real(c_double) :: beta = 0.d0                     ! :403; > 0: split_p_grad / grad1_p_update
    logical :: inline_q = .false.                     ! :474; the tracers ride inside d_sw (sw_core.F90:1020-1043)
    logical :: remap_te = .false.                     ! :399; the remap carries total energy (fv_mapz.F90:232-286, :348-360, :576-619)
    ! thermostruct%use_cond / moist_kappa (nonhydrostatic): q_con transported by d_sw and taken out of the Riemann solvers' pm2,
    ! per-cell cappa in the solvers, the heating and the remap; moist: what moist_cv needs (fv3_set_moist)
    logical :: use_cond = .false., moist_kappa = .false.
    type(fv3_moist_params) :: moist
    logical :: convert_ke = .false.
    ! fast_tau_w_sec > 1e-5: Rayleigh damping of w inside SIM1 / SIM (nh_utils.F90:356-367, :1363-1371); RF_fast .and. tau > 0: Ray_fast at
    ! the end of every acoustic substep (dyn_core.F90:1057-1060, :2485-2601); ks: the levels of pure pressure (Ray_fast's k_rf)
    real(c_double) :: fast_tau_w_sec = 0.d0, tau = 0.d0, rf_cutoff = 30.d2
    logical :: RF_fast = .false.
    integer :: ks = 0
    ! flagstruct%do_diss_est (the SKEB dissipation estimate): d_sw returns diss_e of every level, the loop sums it into diss_est
    ! over the acoustic substeps (dyn_core.F90:805-811); also a member of the gridstruct the context uploads (fv3_domain%do_diss_est)
    logical :: do_diss_est = .false.
    ! flagstruct%fill_dp: mix_dp after d_sw (dyn_core.F90:820, :2119-2200) -- layers thinner than 1 % of their reference thickness take
    ! mass (and the w / pt that goes with it) from a neighbour; the reference thickness is that of the ak / bk given to fv3_host_init_grid
    logical :: fill_dp = .false.
  end type

  !> device-resident state and work arrays of one rank (fv_atmos_type members + dyn_core.F90:256-283)
  type fv3_atmos
    type(c_ptr) :: ctx = c_null_ptr
    type(fv3_flags) :: fl
    integer :: is, ie, js, je, isd, ied, jsd, jed, npz, nq, nx, ny
    integer(c_size_t) :: nA, nU, nV, nB, nCC, nCX, nCY, nFX, nFY
    type(fv3_nh_consts) :: cn
    real(c_double) :: da_min = 0.d0
    ! prognostic fields and their ping-pong partners
    type(c_ptr) :: u, v, w, delp, pt, u_n, v_n, w_n, delp_n, pt_n
    type(c_ptr) :: delz, phis, zs, q, q_n, dp1, dp1_n
    ! work arrays
    type(c_ptr) :: delpc, ptc, uc, vc, ua, va, omga, ut, vt, divgd, gz, pkc, zh, zh_n, pk3
    type(c_ptr) :: crx, xfx, cry, yfx, mfx, mfy, cx, cy, heat_s, diss_e, pk, ws3, ws, pe, peln, ps, pkz
    type(c_ptr) :: divg2, heat_source                 ! external-mode damping field (A), accumulated heat source (A x npz)
    type(c_ptr) :: diss_est = c_null_ptr              ! do_diss_est: the accumulated dissipation estimate (A x npz), allocated on first use
    type(c_ptr) :: du = c_null_ptr, dv = c_null_ptr   ! beta > 0: the saved hydrostatic pressure gradient (dyn_core.F90:278-283)
    type(c_ptr) :: fx_s = c_null_ptr, fy_s = c_null_ptr ! inline_q: the delp fluxes of one substep (FX / FY x npz)
    type(c_ptr) :: q_con = c_null_ptr, q_con_n = c_null_ptr, cappa = c_null_ptr   ! use_cond / moist_kappa (A x npz)
    real(c_double), allocatable :: ak(:), bk(:)
    real(c_double), allocatable :: pfull(:)           ! the caller's pfull (dyn_core's argument) for fast_tau_w_sec / Ray_fast; not set: fv_dynamics.F90:254-262
    ! several ranks of a doubly periodic px x py layout (fv3_host_comm_layout): the neighbour ranks of the eight directions
    integer :: nranks = 1
    integer(c_int) :: peers_to(8) = 0_c_int, peers_from(8) = 0_c_int
    ! fv3_fv_dynamics_call: the energy fixer's columns, g_sum's weights, the Rayleigh profile, u2f of Rayleigh_Friction
    type(c_ptr) :: te0 = c_null_ptr, te = c_null_ptr, zs1 = c_null_ptr, zs0 = c_null_ptr, u2f = c_null_ptr
    real(c_double), allocatable :: area(:,:), rf(:), pm(:)
    integer :: kmax = -1
    real(c_double) :: e_flux = 0.d0, dtmp = 0.d0
    ! flagstruct%consv_am (fv_dynamics.F90:358-361, :747-800; fv3_host_set_consv_am): cos(agrid(:,:,2)) (A), gridstruct%l2c_u / l2c_v
    ! (U / V, zero outside the compute domain) on the device, idiag%zxg of the compute domain on the host; teq, aam, m_fac (CC), ps2 (A)
    logical :: consv_am = .false.
    type(c_ptr) :: am_coslat = c_null_ptr, am_l2c_u = c_null_ptr, am_l2c_v = c_null_ptr
    type(c_ptr) :: am_teq = c_null_ptr, am_aam = c_null_ptr, am_mfac = c_null_ptr, am_ps2 = c_null_ptr
    real(c_double), allocatable :: am_zxg(:,:)
    real(c_double) :: am_omega = 7.292d-5, u00 = 0.d0
    logical :: rfw_ready = .false., rff_ready = .false.   ! RFw_initialized (nh_utils.F90:54), RFF_initialized (dyn_core.F90:84)
  end type

  logical, save :: host_comm = .false.

contains

  subroutine dmalloc(p, n)
    type(c_ptr), intent(out) :: p
    integer(c_size_t), intent(in) :: n
    call fv3_check(fv3_malloc(p, n * 8_c_size_t), 'fv3_malloc')
  end subroutine

  subroutine dzero(at, p, n)
    type(fv3_atmos), intent(in) :: at
    type(c_ptr), intent(in) :: p
    integer(c_size_t), intent(in) :: n
    real(c_double), allocatable, target :: z(:)
    allocate(z(n)); z = 0.d0
    call fv3_check(fv3_memcpy_h2d(at%ctx, p, c_loc(z), n * 8_c_size_t), 'fv3_memcpy_h2d')
    call fv3_check(fv3_sync(at%ctx), 'fv3_sync')
  end subroutine

  subroutine swap(a, b)
    type(c_ptr), intent(inout) :: a, b
    type(c_ptr) :: t
    t = a; a = b; b = t
  end subroutine

  !> group halo update on one rank of the doubly periodic domain (fv_mp_mod.F90:646-876, contacts :473-483).
  !> host_comm (environment FV3_HOST_COMM=1): through the exchange behind the C ABI -- fv3_halo_start posts the eight
  !> messages of the group on the context's RCCL communicator (every neighbour of the single rank is the rank itself),
  !> fv3_halo_complete waits and unpacks -- which is the form the several-rank host uses (to / from = the neighbour ranks).
  subroutine halo(at, field, kind, nk)
    type(fv3_atmos), intent(in) :: at
    type(c_ptr), intent(in) :: field
    integer(c_int), intent(in) :: kind
    integer, intent(in) :: nk
    type(fv3_halo_field) :: hf(1)
    integer(c_int) :: peers(8)
    if (host_comm .or. at%nranks > 1) then
      hf(1)%field = field; hf(1)%kind = kind; hf(1)%nk = int(nk, c_int)
      peers = 0_c_int
      if (at%nranks > 1) then
        call fv3_check(fv3_halo_start(at%ctx, 1_c_int, hf, at%peers_to, at%peers_from), 'fv3_halo_start')
      else
        call fv3_check(fv3_halo_start(at%ctx, 1_c_int, hf, peers, peers), 'fv3_halo_start')
      end if
      call fv3_check(fv3_halo_complete(at%ctx), 'fv3_halo_complete')
    else
      call fv3_check(fv3_halo_fill_periodic(at%ctx, field, kind, int(nk, c_int)), 'fv3_halo_fill_periodic')
    end if
  end subroutine

  !> the same for callers outside this module (kind: 0 = A, 1 = U, 2 = V, 3 = B)
  subroutine fv3_host_halo(at, field, kind, nk)
    type(fv3_atmos), intent(in) :: at
    type(c_ptr), intent(in) :: field
    integer, intent(in) :: kind, nk
    call halo(at, field, int(kind, c_int), nk)
  end subroutine

  !> several ranks of a doubly periodic px x py layout (tools/fv_mp_mod.F90:276-392 with io_layout 1 x 1, contacts :473-483): rank r
  !> holds block (mod(r, px), r / px); the communicator of the exchange behind the C ABI (id: rank 0's fv3_comm_get_unique_id, broadcast
  !> by the caller) and the neighbour ranks of the eight directions in the order of fv3_halo_message_elems (dj = -1, 0, 1 outer,
  !> di = -1, 0, 1 inner): a message to direction d is received from direction -d
  subroutine fv3_host_comm_layout(at, rank, nranks, px, py, id)
    type(fv3_atmos), intent(inout) :: at
    integer, intent(in) :: rank, nranks, px, py
    integer(c_signed_char), intent(in) :: id(128)
    integer :: di, dj, n, ix, iy
    if (px * py /= nranks) error stop 'fv3_host_comm_layout: px * py must be the number of ranks'
    call fv3_check(fv3_comm_init(at%ctx, int(rank, c_int), int(nranks, c_int), id), 'fv3_comm_init')
    ix = mod(rank, px); iy = rank / px
    n = 0
    do dj = -1, 1
      do di = -1, 1
        if (di == 0 .and. dj == 0) cycle
        n = n + 1
        at%peers_to(n) = int(modulo(iy + dj, py) * px + modulo(ix + di, px), c_int)
        at%peers_from(n) = int(modulo(iy - dj, py) * px + modulo(ix - di, px), c_int)
      end do
    end do
    at%nranks = nranks
  end subroutine

  !> the communicator of the exchange behind the C ABI: rank 0 makes the id, a several-rank host broadcasts it (MPI_Bcast)
  subroutine host_comm_init(at)
    type(fv3_atmos), intent(in) :: at
    integer(c_signed_char) :: id(128)
    character(len=8) :: v
    integer :: st
    call get_environment_variable('FV3_HOST_COMM', v, status=st)
    host_comm = (st == 0 .and. v(1:1) == '1')
    if (.not. host_comm) return
    call fv3_check(fv3_comm_get_unique_id(id), 'fv3_comm_get_unique_id')
    call fv3_check(fv3_comm_init(at%ctx, 0_c_int, 1_c_int, id), 'fv3_comm_init')
    print '(a)', ' fv3_host: group halo updates through fv3_halo_start / fv3_halo_complete'
  end subroutine

  !> Create the context, build and upload the gridstruct of the doubly periodic Cartesian tile (what
  !> fv_grid_tools.F90:1202-1221 and fv_grid_utils.F90:426-437,614-683 set for grid_type = 4), allocate the device
  !> arrays, upload the per-level damping coefficients (dyn_core.F90:666-733), dp_ref (:241-244) and ak, bk.
  subroutine fv3_host_init(at, nx, ny, npz, nq, dx_const, dy_const, f0_const, fl, ak, bk)
    type(fv3_atmos), intent(out) :: at
    integer, intent(in) :: nx, ny, npz, nq
    real(c_double), intent(in) :: dx_const, dy_const, f0_const
    type(fv3_flags), intent(in) :: fl
    real(c_double), intent(in) :: ak(npz+1), bk(npz+1)
    type(fv3_domain) :: dom
    type(fv3_grid_host) :: gh
    integer(c_size_t) :: n1, nA
    real(c_double), allocatable, target :: m_dx(:), m_dy(:), m_rdx(:), m_rdy(:), m_area(:), m_rarea(:), m_one(:), m_zero(:), &
                                           m_f0(:), m_sg(:), m_cg(:), r_u(:), r6_u(:), r_v(:), r6_v(:)

    dom%is = 1; dom%ie = nx; dom%js = 1; dom%je = ny; dom%ng = NG
    dom%npx = nx + 1; dom%npy = ny + 1; dom%npz = npz; dom%grid_type = 4
    dom%do_diss_est = 0; dom%prevent_diss_cooling = 1; dom%stretched_grid = 0; dom%lim_fac = 1.d0

    ! ---- gridstruct: every metric term is a constant on this domain; one host array per distinct value is enough,
    !      sized for the largest stagger (B) -- fv3_grid_upload copies the leading part it needs
    nA = int(nx + 2*NG, c_size_t) * (ny + 2*NG)
    n1 = int(nx + 2*NG + 1, c_size_t) * (ny + 2*NG + 1)
    allocate(m_dx(n1), m_dy(n1), m_rdx(n1), m_rdy(n1), m_area(n1), m_rarea(n1), m_one(n1), m_zero(n1), m_f0(n1))
    allocate(m_sg(9 * nA), m_cg(9 * nA))
    m_dx = dx_const; m_dy = dy_const; m_rdx = 1.d0 / dx_const; m_rdy = 1.d0 / dy_const
    m_area = dx_const * dy_const; m_rarea = 1.d0 / (dx_const * dy_const)
    m_one = 1.d0; m_zero = 0.d0; m_f0 = f0_const; m_sg = 1.d0; m_cg = 0.d0
    gh%da_min = dx_const * dy_const; gh%da_min_c = dx_const * dy_const
    gh%area = c_loc(m_area);   gh%rarea = c_loc(m_rarea)
    gh%dxa = c_loc(m_dx);      gh%dya = c_loc(m_dy);     gh%rdxa = c_loc(m_rdx);  gh%rdya = c_loc(m_rdy)
    gh%cosa_s = c_loc(m_zero); gh%rsin2 = c_loc(m_one);  gh%f0 = c_loc(m_f0)
    gh%dx = c_loc(m_dx);       gh%rdx = c_loc(m_rdx);    gh%dyc = c_loc(m_dy);    gh%rdyc = c_loc(m_rdy)
    gh%cosa_v = c_loc(m_zero); gh%sina_v = c_loc(m_one); gh%rsin_v = c_loc(m_one)
    gh%dy = c_loc(m_dy);       gh%rdy = c_loc(m_rdy);    gh%dxc = c_loc(m_dx);    gh%rdxc = c_loc(m_rdx)
    gh%cosa_u = c_loc(m_zero); gh%sina_u = c_loc(m_one); gh%rsin_u = c_loc(m_one)
    ! divg_u = sina_v*dyc/dx, del6_u = sina_v*dx/dyc, divg_v = sina_u*dxc/dy, del6_v = sina_u*dy/dxc (fv_grid_utils.F90:656-665)
    allocate(r_u(n1), r6_u(n1), r_v(n1), r6_v(n1))
    r_u = 1.d0 * dy_const / dx_const;  r6_u = 1.d0 * dx_const / dy_const
    r_v = 1.d0 * dx_const / dy_const;  r6_v = 1.d0 * dy_const / dx_const
    gh%divg_u = c_loc(r_u); gh%del6_u = c_loc(r6_u); gh%divg_v = c_loc(r_v); gh%del6_v = c_loc(r6_v)
    gh%rarea_c = c_loc(m_rarea); gh%fC = c_loc(m_f0); gh%cosa = c_loc(m_zero); gh%sina = c_loc(m_one)
    gh%sin_sg = c_loc(m_sg);     gh%cos_sg = c_loc(m_cg)
    call fv3_host_init_grid(at, dom, gh, nq, fl, ak, bk)
  end subroutine

  !> the general form: bounds / flags of `dom` and the gridstruct members behind `gh` (host addresses with the reference's
  !> shapes, fv_arrays.F90:1749-1881) as the caller holds them -- what fv3_dyn_core_mod's dyn_core hands over
  subroutine fv3_host_init_grid(at, dom, gh, nq, fl, ak, bk)
    type(fv3_atmos), intent(inout) :: at
    type(fv3_domain), intent(in) :: dom
    type(fv3_grid_host), intent(in) :: gh
    integer, intent(in) :: nq
    type(fv3_flags), intent(in) :: fl
    real(c_double), intent(in) :: ak(dom%npz+1), bk(dom%npz+1)
    integer :: nid, njd, k, nx, ny, npz
    integer(c_size_t) :: nk, nk1
    real(c_double), allocatable :: dp_ref(:)

    npz = dom%npz; nx = dom%ie - dom%is + 1; ny = dom%je - dom%js + 1
    at%fl = fl
    at%is = dom%is; at%ie = dom%ie; at%js = dom%js; at%je = dom%je
    at%isd = dom%is - NG; at%ied = dom%ie + NG; at%jsd = dom%js - NG; at%jed = dom%je + NG
    at%npz = npz; at%nq = nq; at%nx = nx; at%ny = ny
    nid = nx + 2*NG; njd = ny + 2*NG
    at%nA = int(nid, c_size_t) * njd;        at%nU = int(nid, c_size_t) * (njd + 1)
    at%nV = int(nid + 1, c_size_t) * njd;    at%nB = int(nid + 1, c_size_t) * (njd + 1)
    at%nCC = int(nx, c_size_t) * ny;         at%nCX = int(nx + 1, c_size_t) * njd
    at%nCY = int(nid, c_size_t) * (ny + 1);  at%nFX = int(nx + 1, c_size_t) * ny
    at%nFY = int(nx, c_size_t) * (ny + 1)
    if (allocated(at%ak)) deallocate(at%ak, at%bk)
    allocate(at%ak(npz+1), at%bk(npz+1)); at%ak = ak; at%bk = bk
    at%da_min = gh%da_min
    if (.not. c_associated(at%ctx)) call fv3_check(fv3_create(dom, at%ctx), 'fv3_create')
    call fv3_check(fv3_grid_upload(at%ctx, gh), 'fv3_grid_upload')
    block      ! area of the compute domain: the weights of g_sum (fv_mapz.F90:736)
      real(c_double), pointer :: a(:,:)
      if (allocated(at%area)) deallocate(at%area)
      allocate(at%area(nx, ny))
      call c_f_pointer(gh%area, a, [nx + 2*NG, ny + 2*NG])
      at%area = a(NG+1:NG+nx, NG+1:NG+ny)
    end block
    call host_comm_init(at)

    ! ---- device arrays ----
    nk = int(npz, c_size_t); nk1 = nk + 1
    call dmalloc(at%u, at%nU*nk);     call dmalloc(at%u_n, at%nU*nk)
    call dmalloc(at%v, at%nV*nk);     call dmalloc(at%v_n, at%nV*nk)
    call dmalloc(at%w, at%nA*nk);     call dmalloc(at%w_n, at%nA*nk)
    call dmalloc(at%delp, at%nA*nk);  call dmalloc(at%delp_n, at%nA*nk)
    call dmalloc(at%pt, at%nA*nk);    call dmalloc(at%pt_n, at%nA*nk)
    call dmalloc(at%delz, at%nCC*nk); call dmalloc(at%phis, at%nA);       call dmalloc(at%zs, at%nA)
    call dmalloc(at%delpc, at%nA*nk); call dmalloc(at%ptc, at%nA*nk)
    call dmalloc(at%uc, at%nV*nk);    call dmalloc(at%vc, at%nU*nk)
    call dmalloc(at%ua, at%nA*nk);    call dmalloc(at%va, at%nA*nk);      call dmalloc(at%omga, at%nA*nk)
    call dmalloc(at%ut, at%nA*nk);    call dmalloc(at%vt, at%nA*nk);      call dmalloc(at%divgd, at%nB*nk)
    call dmalloc(at%gz, at%nA*nk1);   call dmalloc(at%pkc, at%nA*nk1)
    call dmalloc(at%zh, at%nA*nk1);   call dmalloc(at%zh_n, at%nA*nk1);   call dmalloc(at%pk3, at%nA*nk1)
    call dmalloc(at%crx, at%nCX*nk);  call dmalloc(at%xfx, at%nCX*nk)
    call dmalloc(at%cry, at%nCY*nk);  call dmalloc(at%yfx, at%nCY*nk)
    call dmalloc(at%mfx, at%nFX*nk);  call dmalloc(at%mfy, at%nFY*nk)
    call dmalloc(at%cx, at%nCX*nk);   call dmalloc(at%cy, at%nCY*nk)
    call dmalloc(at%heat_s, at%nCC*nk); call dmalloc(at%diss_e, at%nCC*nk); call dmalloc(at%pk, at%nCC*nk1)
    call dmalloc(at%ws3, at%nA);      call dmalloc(at%ws, at%nCC)
    call dmalloc(at%pe, int(nx+2, c_size_t)*nk1*(ny+2));  call dmalloc(at%peln, int(nx, c_size_t)*nk1*ny)
    call dmalloc(at%ps, at%nA);       call dmalloc(at%pkz, at%nCC*nk)
    call dmalloc(at%divg2, at%nA);    call dmalloc(at%heat_source, at%nA*nk)
    call dzero(at, at%divg2, at%nA);  call dzero(at, at%heat_source, at%nA*nk); call dzero(at, at%pkz, at%nCC*nk)
    call dmalloc(at%dp1, at%nA*nk);   call dmalloc(at%dp1_n, at%nA*nk)
    if (fl%inline_q .and. nq > 0) then
      call dmalloc(at%fx_s, at%nFX*nk); call dmalloc(at%fy_s, at%nFY*nk)
    end if
    if (fl%use_cond .or. fl%moist_kappa) then
      if (fl%hydrostatic) error stop 'fv3_host_mod: use_cond / moist_kappa are nonhydrostatic branches'
      call dmalloc(at%q_con, at%nA*nk); call dmalloc(at%q_con_n, at%nA*nk); call dmalloc(at%cappa, at%nA*nk)
      call dzero(at, at%q_con, at%nA*nk); call dzero(at, at%q_con_n, at%nA*nk); call dzero(at, at%cappa, at%nA*nk)
    end if
    if (fl%beta < 0.d0 .and. (fl%hydrostatic .or. fl%beta >= -0.1d0)) &
      error stop 'fv3_host_mod: beta < 0: only beta < -0.1 in the nonhydrostatic loop selects anything (one_grad_p, dyn_core.F90:1029)'
    if (fl%beta > 1.d-9) then
      call dmalloc(at%du, at%nU*nk); call dmalloc(at%dv, at%nV*nk)
      call dzero(at, at%du, at%nU*nk); call dzero(at, at%dv, at%nV*nk)
    end if
    at%q = c_null_ptr; at%q_n = c_null_ptr
    if (nq > 0) then
      call dmalloc(at%q, at%nA*nk*nq); call dmalloc(at%q_n, at%nA*nk*nq)
      call dzero(at, at%q_n, at%nA*nk*nq)
    end if
    ! the arrays the kernels read before anything writes them (halos, accumulators) start from zero like ctx.zeros
    call dzero(at, at%u_n, at%nU*nk);   call dzero(at, at%v_n, at%nV*nk);  call dzero(at, at%w_n, at%nA*nk)
    call dzero(at, at%delp_n, at%nA*nk); call dzero(at, at%pt_n, at%nA*nk)
    call dzero(at, at%delpc, at%nA*nk); call dzero(at, at%ptc, at%nA*nk)
    call dzero(at, at%uc, at%nV*nk);    call dzero(at, at%vc, at%nU*nk)
    call dzero(at, at%ua, at%nA*nk);    call dzero(at, at%va, at%nA*nk);   call dzero(at, at%omga, at%nA*nk)
    call dzero(at, at%ut, at%nA*nk);    call dzero(at, at%vt, at%nA*nk);   call dzero(at, at%divgd, at%nB*nk)
    call dzero(at, at%gz, at%nA*nk1);   call dzero(at, at%pkc, at%nA*nk1)
    call dzero(at, at%zh, at%nA*nk1);   call dzero(at, at%zh_n, at%nA*nk1); call dzero(at, at%pk3, at%nA*nk1)
    call dzero(at, at%crx, at%nCX*nk);  call dzero(at, at%xfx, at%nCX*nk)
    call dzero(at, at%cry, at%nCY*nk);  call dzero(at, at%yfx, at%nCY*nk)
    call dzero(at, at%mfx, at%nFX*nk);  call dzero(at, at%mfy, at%nFY*nk)
    call dzero(at, at%cx, at%nCX*nk);   call dzero(at, at%cy, at%nCY*nk)
    call dzero(at, at%heat_s, at%nCC*nk); call dzero(at, at%diss_e, at%nCC*nk); call dzero(at, at%pk, at%nCC*nk1)
    call dzero(at, at%ws3, at%nA);      call dzero(at, at%ws, at%nCC)
    call dzero(at, at%pe, int(nx+2, c_size_t)*nk1*(ny+2)); call dzero(at, at%peln, int(nx, c_size_t)*nk1*ny)
    call dzero(at, at%ps, at%nA);       call dzero(at, at%pkz, at%nCC*nk)
    call dzero(at, at%dp1, at%nA*nk);   call dzero(at, at%dp1_n, at%nA*nk)

    call upload_levels(at)
    allocate(dp_ref(npz))
    do k = 1, npz
      dp_ref(k) = (ak(k+1) - ak(k)) + (bk(k+1) - bk(k)) * 1.d5          ! dyn_core.F90:241-244
    end do
    call fv3_check(fv3_set_dp_ref(at%ctx, dp_ref), 'fv3_set_dp_ref')
    call fv3_check(fv3_set_ak_bk(at%ctx, at%ak, at%bk), 'fv3_set_ak_bk')
    at%cn%grav = fl%grav; at%cn%rdgas = fl%rdgas; at%cn%cp_air = fl%cp_air; at%cn%akap = fl%akap
    at%cn%ptop = fl%ptop; at%cn%p_fac = fl%p_fac; at%cn%a_imp = fl%a_imp; at%cn%m_split = int(max(1, fl%m_split), c_int)
  end subroutine

  !> nord_k, nord_v, nord_w, nord_t, d2_divg, damp_vt, damp_w, damp_t, d_con_k per level: the k loop of dyn_core.F90:666-733
  subroutine upload_levels(at)
    type(fv3_atmos), intent(in) :: at
    integer(c_int), allocatable, target :: nord_k(:), nord_v(:), nord_w(:), nord_t(:)
    real(c_double), allocatable, target :: d2_divg(:), damp_vt(:), damp_w(:), damp_t(:), d_con_k(:)
    type(fv3_dsw_levels) :: lv
    integer :: k, npz
    npz = at%npz
    allocate(nord_k(npz), nord_v(npz), nord_w(npz), nord_t(npz), d2_divg(npz), damp_vt(npz), damp_w(npz), damp_t(npz), &
             d_con_k(npz))
    do k = 1, npz
      nord_k(k) = at%fl%nord
      nord_v(k) = min(2, at%fl%nord)
      d2_divg(k) = min(0.20d0, at%fl%d2_bg)
      damp_vt(k) = 0.d0
      if (at%fl%do_vort_damp) damp_vt(k) = at%fl%vtdm4
      nord_w(k) = nord_v(k); nord_t(k) = nord_v(k); damp_w(k) = damp_vt(k); damp_t(k) = damp_vt(k)
      d_con_k(k) = at%fl%d_con
      if (npz == 1 .or. at%fl%n_sponge < 0) then
        d2_divg(k) = at%fl%d2_bg
      else if (k == 1) then
        nord_k(k) = 0
        if (at%fl%is_ideal_case) then
          d2_divg(k) = max(at%fl%d2_bg, at%fl%d2_bg_k1)
        else
          d2_divg(k) = max(0.01d0, at%fl%d2_bg, at%fl%d2_bg_k1)
        end if
        nord_w(k) = 0; damp_w(k) = d2_divg(k)
        if (at%fl%do_vort_damp) then
          nord_v(k) = 0; damp_vt(k) = 0.5d0 * d2_divg(k)
        end if
        d_con_k(k) = 0.d0
      else if (k == 2 .and. at%fl%d2_bg_k2 > 0.01d0) then
        nord_k(k) = 0
        d2_divg(k) = max(at%fl%d2_bg, at%fl%d2_bg_k2)
        nord_w(k) = 0; damp_w(k) = d2_divg(k)
        if (at%fl%do_vort_damp) then
          nord_v(k) = 0; damp_vt(k) = 0.5d0 * d2_divg(k)
        end if
        d_con_k(k) = 0.d0
      else if (k == 3 .and. at%fl%d2_bg_k2 > 0.05d0) then
        nord_k(k) = 0
        d2_divg(k) = max(at%fl%d2_bg, 0.2d0 * at%fl%d2_bg_k2)
        nord_w(k) = 0; damp_w(k) = d2_divg(k)
        d_con_k(k) = 0.d0
      end if
    end do
    lv%nord_k = c_loc(nord_k); lv%nord_v = c_loc(nord_v); lv%nord_w = c_loc(nord_w); lv%nord_t = c_loc(nord_t)
    lv%d2_divg = c_loc(d2_divg); lv%damp_vt = c_loc(damp_vt); lv%damp_w = c_loc(damp_w); lv%damp_t = c_loc(damp_t)
    lv%d_con_k = c_loc(d_con_k)
    call fv3_check(fv3_dsw_levels_upload(at%ctx, lv), 'fv3_dsw_levels_upload')
  end subroutine

  !> host arrays with the reference's shapes (fv_arrays.F90:1521-1563) -> device
  subroutine fv3_host_upload(at, u, v, w, delp, pt, delz, phis, q)
    type(fv3_atmos), intent(inout) :: at
    real(c_double), intent(in), target :: u(at%isd:at%ied, at%jsd:at%jed+1, at%npz), v(at%isd:at%ied+1, at%jsd:at%jed, at%npz)
    real(c_double), intent(in), target :: w(at%isd:at%ied, at%jsd:at%jed, at%npz), delp(at%isd:at%ied, at%jsd:at%jed, at%npz)
    real(c_double), intent(in), target :: pt(at%isd:at%ied, at%jsd:at%jed, at%npz), delz(at%is:at%ie, at%js:at%je, at%npz)
    real(c_double), intent(in), target :: phis(at%isd:at%ied, at%jsd:at%jed)
    real(c_double), intent(in), target, optional :: q(at%isd:at%ied, at%jsd:at%jed, at%npz, *)
    real(c_double), allocatable, target :: zs(:,:)
    integer(c_size_t) :: nk
    nk = int(at%npz, c_size_t)
    call put(at%u, c_loc(u), at%nU*nk);       call put(at%v, c_loc(v), at%nV*nk);   call put(at%w, c_loc(w), at%nA*nk)
    call put(at%delp, c_loc(delp), at%nA*nk); call put(at%pt, c_loc(pt), at%nA*nk)
    call put(at%delz, c_loc(delz), at%nCC*nk); call put(at%phis, c_loc(phis), at%nA)
    allocate(zs(at%isd:at%ied, at%jsd:at%jed))
    zs = phis * (1.d0 / at%fl%grav)                                       ! dyn_core.F90:246-251
    call put(at%zs, c_loc(zs), at%nA)
    if (present(q) .and. at%nq > 0) call put(at%q, c_loc(q), at%nA*nk*at%nq)
    call fv3_check(fv3_sync(at%ctx), 'fv3_sync')
  contains
    subroutine put(d, h, n)
      type(c_ptr), intent(in) :: d, h
      integer(c_size_t), intent(in) :: n
      call fv3_check(fv3_memcpy_h2d(at%ctx, d, h, n * 8_c_size_t), 'fv3_memcpy_h2d')
    end subroutine
  end subroutine

  subroutine fv3_host_download(at, u, v, w, delp, pt, delz, q, ua, va)
    type(fv3_atmos), intent(in) :: at
    real(c_double), intent(out), target :: u(at%isd:at%ied, at%jsd:at%jed+1, at%npz), v(at%isd:at%ied+1, at%jsd:at%jed, at%npz)
    real(c_double), intent(out), target :: w(at%isd:at%ied, at%jsd:at%jed, at%npz), delp(at%isd:at%ied, at%jsd:at%jed, at%npz)
    real(c_double), intent(out), target :: pt(at%isd:at%ied, at%jsd:at%jed, at%npz), delz(at%is:at%ie, at%js:at%je, at%npz)
    real(c_double), intent(out), target, optional :: q(at%isd:at%ied, at%jsd:at%jed, at%npz, *)
    real(c_double), intent(out), target, optional :: ua(at%isd:at%ied, at%jsd:at%jed, at%npz), va(at%isd:at%ied, at%jsd:at%jed, at%npz)
    integer(c_size_t) :: nk
    nk = int(at%npz, c_size_t)
    call fv3_check(fv3_sync(at%ctx), 'fv3_sync')
    call get(c_loc(u), at%u, at%nU*nk);       call get(c_loc(v), at%v, at%nV*nk);   call get(c_loc(w), at%w, at%nA*nk)
    call get(c_loc(delp), at%delp, at%nA*nk); call get(c_loc(pt), at%pt, at%nA*nk)
    call get(c_loc(delz), at%delz, at%nCC*nk)
    if (present(q) .and. at%nq > 0) call get(c_loc(q), at%q, at%nA*nk*at%nq)
    if (present(ua)) call get(c_loc(ua), at%ua, at%nA*nk)
    if (present(va)) call get(c_loc(va), at%va, at%nA*nk)
    call fv3_check(fv3_sync(at%ctx), 'fv3_sync')
  contains
    subroutine get(h, d, n)
      type(c_ptr), intent(in) :: d, h
      integer(c_size_t), intent(in) :: n
      call fv3_check(fv3_memcpy_d2h(at%ctx, h, d, n * 8_c_size_t), 'fv3_memcpy_d2h')
    end subroutine
  end subroutine

  !> do_diss_est: the accumulator exists from the first call on, zeroed once (dyn_core.F90:285 zeroes it on init_step only: the
  !> reference-signature wrapper hands its caller's array in and out around every call instead)
  subroutine diss_est_begin(at)
    type(fv3_atmos), intent(inout) :: at
    if (.not. at%fl%do_diss_est) return
    if (c_associated(at%diss_est)) return
    call dmalloc(at%diss_est, at%nA*int(at%npz, c_size_t))
    call dzero(at, at%diss_est, at%nA*int(at%npz, c_size_t))
  end subroutine

  !> the acoustic substep loop, nonhydrostatic branch (dyn_core.F90:313-1286)
  subroutine fv3_dyn_core(at, bdt)
    type(fv3_atmos), intent(inout) :: at
    real(c_double), intent(in) :: bdt
    type(fv3_dsw_params) :: par
    real(c_double) :: dt, dt2, rdt, ptk, peln1, top
    integer :: it, n_split, npz
    logical :: remap_step
    integer(c_int) :: last_call, use_logp
    type(c_ptr) :: ctx, fxp, fyp, qcp, qcn, dv2n, hsp, dsp
    logical :: heating
    integer :: n_con
    if (at%fl%hydrostatic) then
      call fv3_dyn_core_hydrostatic(at, bdt)
      return
    end if
    ctx = at%ctx; npz = at%npz
    n_split = at%fl%n_split
    dt = bdt / real(n_split, c_double)
    dt2 = 0.5d0 * dt
    rdt = 1.d0 / dt
    heating = at%fl%d_con > 1.d-5                                         ! dyn_core.F90:294
    if (heating) call dzero(at, at%heat_source, at%nA*npz)
    call diss_est_begin(at)
    ptk = at%fl%ptop ** at%fl%akap                                        ! dyn_core.F90:222
    peln1 = log(at%fl%ptop)
    use_logp = merge(1_c_int, 0_c_int, at%fl%use_logp)
    top = merge(peln1, ptk, at%fl%use_logp)
    call dzero(at, at%mfx, at%nFX*npz); call dzero(at, at%mfy, at%nFY*npz)        ! :289-292
    call dzero(at, at%cx, at%nCX*npz);  call dzero(at, at%cy, at%nCY*npz)
    par%dt = dt; par%hord_tr = at%fl%hord_tr; par%hord_mt = at%fl%hord_mt; par%hord_vt = at%fl%hord_vt
    par%hord_tm = at%fl%hord_tm; par%hord_dp = at%fl%hord_dp; par%dddmp = at%fl%dddmp; par%d4_bg = at%fl%d4_bg
    par%kgb = at%fl%ke_bg; par%hydrostatic = 0; par%use_cond = merge(1_c_int, 0_c_int, at%fl%use_cond)
    ! fv_dynamics.F90:467-470: halo of delp, pt (pack 1) and u, v (pack 8) before the first substep
    call halo(at, at%delp, KIND_A, npz); call halo(at, at%pt, KIND_A, npz)
    call halo(at, at%u, KIND_U, npz);    call halo(at, at%v, KIND_V, npz)
    do it = 1, n_split
      remap_step = it == n_split
      last_call = merge(1_c_int, 0_c_int, remap_step)
      call halo(at, at%w, KIND_A, npz)                                                    ! :350 / :432 (pack 7)
      if (it == 1) then                                                                   ! :353-389
        call fv3_check(fv3_zh_from_delz(ctx, at%zs, at%delz, at%zh), 'zh_from_delz')
        call halo(at, at%zh, KIND_A, npz + 1)                                             ! gz halo (pack 5); zh = gz (:491-499)
      end if
      call fv3_check(fv3_c_sw(ctx, at%delpc, at%delp, at%ptc, at%pt, at%u, at%v, at%w, at%uc, at%vc, at%ua, at%va, &
                              at%omga, at%ut, at%vt, at%divgd, int(at%fl%nord, c_int), dt2, 0_c_int, 1_c_int), 'c_sw')  ! :439-447
      if (at%fl%nord > 0) call halo(at, at%divgd, KIND_B, npz)                            ! :451 / :577 (pack 3, CORNER)
      call fv3_check(fv3_update_dz_c(ctx, dt2, at%zs, at%ut, at%vt, at%zh, at%gz, at%ws3), 'update_dz_c')   ! :514-527
      call set_condensate(at)
      call host_fast_tau_w(at, dt2)
      call fv3_check(fv3_riem_solver_c(ctx, dt2, at%cn, at%phis, at%omga, at%ptc, at%delpc, at%gz, at%pkc, at%ws3), &
                     'riem_solver_c')                                                     ! :531
      call fv3_check(fv3_p_grad_c(ctx, dt2, at%delpc, at%pkc, at%gz, at%uc, at%vc, 0_c_int), 'p_grad_c')    ! :562
      call halo(at, at%uc, KIND_V, npz); call halo(at, at%vc, KIND_U, npz)                ! :565 / :578 (pack 9, CGRID_NE)
      call inline_q_begin(at, fxp, fyp)
      qcp = c_null_ptr; qcn = c_null_ptr
      if (at%fl%use_cond) then
        qcp = at%q_con; qcn = at%q_con_n
      end if
      hsp = c_null_ptr; dsp = c_null_ptr    ! heat_s is read only when d_con > 1e-5 (:798-803), diss_e with do_diss_est (:805-811)
      if (heating) hsp = at%heat_s
      if (at%fl%do_diss_est) dsp = at%diss_e
      call fv3_check(fv3_d_sw(ctx, par, at%vt, at%delp, at%pt, at%u, at%v, at%w, at%uc, at%vc, at%ua, at%va, at%divgd, &
                              fxp, fyp, at%cx, at%cy, at%crx, at%cry, at%xfx, at%yfx, qcp, &
                              at%delp_n, at%pt_n, at%u_n, at%v_n, at%w_n, qcn, hsp, dsp), 'd_sw')  ! :762
      if (heating) call fv3_check(fv3_heat_source_accum(ctx, at%heat_source, at%heat_s), 'heat_source_accum')   ! :798-803
      if (at%fl%do_diss_est) call fv3_check(fv3_heat_source_accum(ctx, at%diss_est, at%diss_e), 'diss_est += diss_e')   ! :805-811
      call inline_q_end(at)
      ! beta < -0.1: the external-mode damping field of one_grad_p (:1030) from the delp before d_sw and d_sw's divergence (:745-747, :791-848)
      if (at%fl%beta < -0.1d0 .and. at%fl%d_ext > 0.d0) &
        call fv3_check(fv3_divg2_ext(ctx, at%fl%d_ext, at%delp, at%vt, at%divg2), 'divg2_ext')
      call swap(at%delp, at%delp_n); call swap(at%pt, at%pt_n)
      call swap(at%u, at%u_n); call swap(at%v, at%v_n); call swap(at%w, at%w_n)
      if (at%fl%fill_dp) call fv3_check(fv3_mix_dp(ctx, 0_c_int, at%w, at%delp, at%pt), 'mix_dp')    ! :820
      call halo(at, at%delp, KIND_A, npz); call halo(at, at%pt, KIND_A, npz)              ! :823-824 / :851 (pack 1)
      if (at%fl%use_cond) then
        call swap(at%q_con, at%q_con_n)
        call halo(at, at%q_con, KIND_A, npz)                                              ! :825 / :852 (pack 11)
        call set_condensate(at)
      end if
      call fv3_check(fv3_update_dz_d(ctx, int(at%fl%hord_tm, c_int), at%zs, at%zh, at%zh_n, at%crx, at%cry, at%xfx, &
                                     at%yfx, at%ws, rdt), 'update_dz_d')                  ! :911
      call swap(at%zh, at%zh_n)
      call fv3_check(fv3_riem_solver3(ctx, dt, at%cn, at%zs, at%w, at%delz, at%pt, at%delp, at%zh, at%pe, at%pkc, &
                                      at%pk3, at%pk, at%peln, at%ws, use_logp, last_call, &
                                      merge(1_c_int, 0_c_int, at%fl%beta < -0.1d0)), 'riem_solver3')  ! :932, fp_out :939
      call halo(at, at%zh, KIND_A, npz + 1); call halo(at, at%pkc, KIND_A, npz + 1)       ! :944-950 (packs 4, 5)
      if (remap_step) call fv3_check(fv3_pe_halo(ctx, at%fl%ptop, at%pe, at%delp), 'pe_halo')   ! :952-953
      call fv3_check(fv3_pk3_halo(ctx, at%fl%ptop, at%fl%akap, at%pk3, at%delp, use_logp), 'pk3_halo')   ! :955-959
      ! :982-989 gz = zh*grav is fused into nh_p_grad (gz_scale)
      if (at%fl%beta > 0.d0) then     ! :1027-1028; beta_d = 0 in the first substep (:398-406)
        call fv3_check(fv3_split_p_grad(ctx, at%u, at%v, at%pkc, at%zh, at%fl%grav, at%delp, at%pk3, merge(0.d0, at%fl%beta, it == 1), &
                                        dt, top, at%du, at%dv), 'split_p_grad')
      else if (at%fl%beta < -0.1d0) then   ! :1029-1030: pkc is the full pressure, the layer weights a2b_ord4 of delp
        dv2n = c_null_ptr
        if (at%fl%d_ext > 0.d0) dv2n = at%divg2
        call fv3_check(fv3_one_grad_p_nh(ctx, at%u, at%v, at%pkc, at%zh, dv2n, at%delp, dt, at%fl%ptop, at%fl%grav), 'one_grad_p (nh)')
      else
        call fv3_check(fv3_nh_p_grad(ctx, at%u, at%v, at%pkc, at%zh, at%fl%grav, at%delp, at%pk3, dt, top), 'nh_p_grad')  ! :1032
      end if
      call host_ray_fast(at, dt)                                                          ! :1057-1060
      if (it /= n_split) then
        call halo(at, at%u, KIND_U, npz); call halo(at, at%v, KIND_V, npz)                ! :1168-1169 (pack 8)
      else if (at%fl%use_old_omega) then
        ! :1182-1191: omga = (pe - pem)*rdt; pem from the delp this substep started with (:409-421) = delp_n after the swap
        call fv3_check(fv3_omga_update(ctx, rdt, at%fl%ptop, at%pe, at%delp_n, at%omga), 'omga_update')
      end if
    end do
    ! dissipative heating (:296-308, :1300-1355)
    n_con = host_n_con(at%fl, npz)
    if (n_con /= 0 .and. heating) then
      call halo(at, at%heat_source, KIND_A, npz)                                          ! del2_cubed's mpp_update_domains, :2399
      call fv3_check(fv3_del2_cubed(ctx, at%heat_source, int(npz, c_int), 0.20d0 * at%da_min, &
                                    int(min(3, at%fl%nord + 1), c_int)), 'del2_cubed')     ! :1301-1303
      call fv3_check(fv3_apply_heat_source(ctx, int(n_con, c_int), 0_c_int, bdt, at%fl%delt_max, at%fl%cp_air, &
                                           at%fl%cp_air - at%fl%rdgas, at%fl%rdgas, at%fl%grav, at%pt, at%heat_source, &
                                           at%delp, at%delz, at%pkz), 'apply_heat_source')
    end if
  end subroutine

  !> number of levels that receive the dissipative heating (dyn_core.F90:296-308)
  integer function host_n_con(fl, npz) result(n_con)
    type(fv3_flags), intent(in) :: fl
    integer, intent(in) :: npz
    if (fl%convert_ke .or. (fl%do_vort_damp .and. fl%vtdm4 > 1.d-4)) then
      n_con = npz
    else if (fl%d2_bg_k1 < 1.d-3) then
      n_con = 0
    else if (fl%d2_bg_k2 < 1.d-3) then
      n_con = 1
    else
      n_con = 2
    end if
  end function

  !> the substep loop with hydrostatic = .true., (dyn_core.F90:313-1286): geopk on the C and D grids (:480-482,
  !> :905-907), p_grad_c (:562), the external-mode damping field (:745-747, :791-848), pk = pkc on the last substep
  !> (:1001-1010), one_grad_p (:1021); the heating of pt afterwards with the hydrostatic pkz
  subroutine fv3_dyn_core_hydrostatic(at, bdt)
    type(fv3_atmos), intent(inout) :: at
    real(c_double), intent(in) :: bdt
    type(fv3_dsw_params) :: par
    real(c_double) :: dt, dt2, ptk
    integer :: it, n_split, npz, n_con
    logical :: heating
    type(c_ptr) :: ctx, dv2, fxp, fyp
    ctx = at%ctx; npz = at%npz
    n_split = at%fl%n_split
    dt = bdt / real(n_split, c_double)
    dt2 = 0.5d0 * dt
    ptk = at%fl%ptop ** at%fl%akap
    heating = at%fl%d_con > 1.d-5
    call dzero(at, at%mfx, at%nFX*npz); call dzero(at, at%mfy, at%nFY*npz)
    call dzero(at, at%cx, at%nCX*npz);  call dzero(at, at%cy, at%nCY*npz)
    if (heating) call dzero(at, at%heat_source, at%nA*npz)
    call diss_est_begin(at)
    par%dt = dt; par%hord_tr = at%fl%hord_tr; par%hord_mt = at%fl%hord_mt; par%hord_vt = at%fl%hord_vt
    par%hord_tm = at%fl%hord_tm; par%hord_dp = at%fl%hord_dp; par%dddmp = at%fl%dddmp; par%d4_bg = at%fl%d4_bg
    par%kgb = at%fl%ke_bg; par%hydrostatic = 1; par%use_cond = 0
    dv2 = c_null_ptr
    if (at%fl%d_ext > 0.d0) dv2 = at%divg2
    call halo(at, at%delp, KIND_A, npz); call halo(at, at%pt, KIND_A, npz)
    call halo(at, at%u, KIND_U, npz);    call halo(at, at%v, KIND_V, npz)
    do it = 1, n_split
      call fv3_check(fv3_c_sw(ctx, at%delpc, at%delp, at%ptc, at%pt, at%u, at%v, c_null_ptr, at%uc, at%vc, at%ua, at%va, &
                              c_null_ptr, at%ut, at%vt, at%divgd, int(at%fl%nord, c_int), dt2, 1_c_int, 1_c_int), 'c_sw')
      if (at%fl%nord > 0) call halo(at, at%divgd, KIND_B, npz)
      call fv3_check(fv3_geopk(ctx, at%fl%ptop, at%fl%akap, at%fl%cp_air, ptk, at%pe, at%peln, at%delpc, at%pkc, at%gz, &
                               at%phis, at%ptc, at%pkz, 1_c_int), 'geopk (C grid)')
      call fv3_check(fv3_p_grad_c(ctx, dt2, at%delpc, at%pkc, at%gz, at%uc, at%vc, 1_c_int), 'p_grad_c')
      call halo(at, at%uc, KIND_V, npz); call halo(at, at%vc, KIND_U, npz)
      call inline_q_begin(at, fxp, fyp)
      call fv3_check(fv3_d_sw(ctx, par, at%vt, at%delp, at%pt, at%u, at%v, c_null_ptr, at%uc, at%vc, at%ua, at%va, at%divgd, &
                              fxp, fyp, at%cx, at%cy, at%crx, at%cry, at%xfx, at%yfx, c_null_ptr, &
                              at%delp_n, at%pt_n, at%u_n, at%v_n, c_null_ptr, c_null_ptr, at%heat_s, at%diss_e), 'd_sw')
      if (heating) call fv3_check(fv3_heat_source_accum(ctx, at%heat_source, at%heat_s), 'heat_source_accum')
      if (at%fl%do_diss_est) call fv3_check(fv3_heat_source_accum(ctx, at%diss_est, at%diss_e), 'diss_est += diss_e')   ! :805-811
      call inline_q_end(at)
      ! the external-mode damping field from the delp BEFORE d_sw (:745-747) and d_sw's divergence output (:791-848)
      call fv3_check(fv3_divg2_ext(ctx, at%fl%d_ext, at%delp, at%vt, at%divg2), 'divg2_ext')
      call swap(at%delp, at%delp_n); call swap(at%pt, at%pt_n); call swap(at%u, at%u_n); call swap(at%v, at%v_n)
      if (at%fl%fill_dp) call fv3_check(fv3_mix_dp(ctx, 1_c_int, c_null_ptr, at%delp, at%pt), 'mix_dp')   ! :820
      call halo(at, at%delp, KIND_A, npz); call halo(at, at%pt, KIND_A, npz)
      call fv3_check(fv3_geopk(ctx, at%fl%ptop, at%fl%akap, at%fl%cp_air, ptk, at%pe, at%peln, at%delp, at%pkc, at%gz, &
                               at%phis, at%pt, at%pkz, 0_c_int), 'geopk')
      if (it == n_split) call fv3_check(fv3_copy_a_to_cc(ctx, at%pkc, at%pk, int(npz + 1, c_int)), 'pk = pkc')
      if (at%fl%beta > 0.d0) then     ! :1018-1019
        call fv3_check(fv3_grad1_p_update(ctx, dv2, at%u, at%v, at%pkc, at%gz, dt, ptk, merge(0.d0, at%fl%beta, it == 1), &
                                          at%du, at%dv), 'grad1_p_update')
      else
        call fv3_check(fv3_one_grad_p(ctx, at%u, at%v, at%pkc, at%gz, dv2, dt, ptk), 'one_grad_p')
      end if
      call host_ray_fast(at, dt)                                                          ! :1057-1060
      if (it /= n_split) then
        call halo(at, at%u, KIND_U, npz); call halo(at, at%v, KIND_V, npz)
      end if
    end do
    n_con = host_n_con(at%fl, npz)
    if (n_con /= 0 .and. heating) then
      call halo(at, at%heat_source, KIND_A, npz)
      call fv3_check(fv3_del2_cubed(ctx, at%heat_source, int(npz, c_int), 0.20d0 * at%da_min, &
                                    int(min(3, at%fl%nord + 1), c_int)), 'del2_cubed')
      call fv3_check(fv3_apply_heat_source(ctx, int(n_con, c_int), 1_c_int, bdt, at%fl%delt_max, at%fl%cp_air, &
                                           at%fl%cp_air - at%fl%rdgas, at%fl%rdgas, at%fl%grav, at%pt, at%heat_source, &
                                           at%delp, c_null_ptr, at%pkz), 'apply_heat_source')
    end if
  end subroutine

  !> tracer_2d (fv_tracer2d.F90:297-557): the host keeps the part with the cross-rank reduction and the integer
  !> bookkeeping (:382-456), the arithmetic runs in the kernels
  subroutine fv3_tracer_2d(at)
    type(fv3_atmos), intent(inout) :: at
    real(c_double), allocatable :: cmax(:), frac(:)
    integer(c_int), allocatable :: ksplt(:)
    real(c_double) :: c_global
    integer :: nsplt, it, npz, k
    npz = at%npz
    allocate(cmax(npz), frac(npz), ksplt(npz))
    call fv3_check(fv3_tracer_2d_prep(at%ctx, int(at%fl%q_split, c_int), at%cx, at%cy, at%xfx, at%yfx, cmax), &
                   'tracer_2d_prep')                                                     ! :362-400
    if (at%fl%q_split == 0) then
      if (at%nranks > 1) call fv3_check(fv3_allreduce_max(at%ctx, cmax, int(npz, c_int)), 'mp_reduce_max')   ! :405
      if (npz /= 1) then                                                                 ! :407-412
        c_global = maxval(cmax)
      else
        c_global = cmax(1)
      end if
      nsplt = int(1.d0 + c_global)
    else
      nsplt = at%fl%q_split
    end if
    if (nsplt /= 1) then                                                                 ! :421-456
      do k = 1, npz
        ksplt(k) = int(1.d0 + cmax(k), c_int)
        frac(k) = 1.d0 / real(ksplt(k), c_double)
      end do
      call fv3_check(fv3_tracer_2d_scale(at%ctx, frac, at%cx, at%xfx, at%mfx, at%cy, at%yfx, at%mfy), 'tracer_2d_scale')
    else
      ksplt = 1
    end if
    if (at%fl%trdm2 > 1.d-4) call halo(at, at%dp1, KIND_A, npz)                           ! dp1_pack, :466
    do it = 1, nsplt                                                                      ! :471-541
      call halo(at, at%q, KIND_A, npz * at%nq)                                            ! q_pack, :474 / :536
      call fv3_check(fv3_tracer_2d_step(at%ctx, int(it, c_int), int(nsplt, c_int), ksplt, int(at%nq, c_int), &
                                        int(at%fl%hord_tr, c_int), int(at%fl%nord_tr, c_int), at%fl%trdm2, at%q, at%q_n, &
                                        at%dp1, at%dp1_n, at%mfx, at%mfy, at%cx, at%cy, at%xfx, at%yfx), 'tracer_2d_step')
      call swap(at%q, at%q_n)
      if (it /= nsplt) call swap(at%dp1, at%dp1_n)
    end do
  end subroutine

  !> one dt_atmos: the k_split loop of fv_dynamics (fv_dynamics.F90:460-665): acoustic substeps, tracer transport,
  !> vertical remap.  pt holds theta_v (what dyn_core works on); last_step makes the final remap return T (fv_mapz.F90:793-821).
  subroutine fv3_fv_dynamics(at, bdt, last_step, last_code)
    type(fv3_atmos), intent(inout) :: at
    real(c_double), intent(in) :: bdt
    logical, intent(in) :: last_step
    integer, intent(in), optional :: last_code      !< what the last remap gets as last_step (2: the energy fixer follows, T_v stays)
    type(fv3_remap_params) :: rp
    integer(c_int), allocatable :: kord_tr(:)
    real(c_double) :: mdt
    integer :: n_map
    mdt = bdt / real(at%fl%k_split, c_double)
    allocate(kord_tr(max(1, at%nq))); kord_tr = int(at%fl%kord_tr, c_int)
    rp%hydrostatic = merge(1_c_int, 0_c_int, at%fl%hydrostatic); rp%adiabatic = merge(1_c_int, 0_c_int, at%fl%adiabatic); rp%nq = int(at%nq, c_int)
    rp%kord_mt = int(at%fl%kord_mt, c_int); rp%kord_wz = int(at%fl%kord_wz, c_int); rp%kord_tm = int(at%fl%kord_tm, c_int)
    rp%sphum = merge(1_c_int, 0_c_int, at%nq > 0); rp%fill = merge(1_c_int, 0_c_int, at%fl%fill)
    rp%akap = at%fl%akap; rp%ptop = at%fl%ptop; rp%rdgas = at%fl%rdgas; rp%grav = at%fl%grav
    rp%cv_air = at%fl%cp_air - at%fl%rdgas; rp%r_vir = at%fl%r_vir; rp%cp = at%fl%cp_air; rp%t_min = at%fl%t_min
    if (at%fl%do_diss_est) then          ! dyn_core zeroes diss_est on init_step = (n_map == 1): once per fv_dynamics call (:497, dyn_core.F90:285)
      call diss_est_begin(at)
      call dzero(at, at%diss_est, at%nA * int(at%npz, c_size_t))
    end if
    do n_map = 1, at%fl%k_split
      call fv3_check(fv3_memcpy_d2d(at%ctx, at%dp1, at%delp, at%nA * at%npz * 8_c_size_t), 'dp1 = delp')   ! :475-481
      if (at%fl%use_cond) call halo(at, at%q_con, KIND_A, at%npz)                                          ! :464 / :487 (pack 11)
      if (at%fl%moist_kappa) call halo(at, at%cappa, KIND_A, at%npz)                                       ! :465 / :488 (pack 12)
      call fv3_dyn_core(at, mdt)                                                                           ! :493
      if (at%nq > 0 .and. .not. at%fl%inline_q) call fv3_tracer_2d(at)                                     ! :509-533
      rp%last_step = merge(1_c_int, 0_c_int, last_step .and. n_map == at%fl%k_split)
      if (present(last_code) .and. last_step .and. n_map == at%fl%k_split) rp%last_step = int(last_code, c_int)
      if (at%fl%remap_te) call fv3_check(fv3_set_remap_te(at%ctx, 1_c_int, at%phis, at%dp1), 'set_remap_te')   ! te = dp1, :612
      if (at%fl%use_cond .or. at%fl%moist_kappa) then          ! q_con is a ping-pong pair: the current buffer
        at%fl%moist%moist_kappa = merge(1_c_int, 0_c_int, at%fl%moist_kappa)
        at%fl%moist%use_cond = merge(1_c_int, 0_c_int, at%fl%use_cond)
        call fv3_check(fv3_set_moist(at%ctx, at%fl%moist, at%q_con, at%cappa), 'set_moist')
      end if
      if (at%fl%hydrostatic) then
        call fv3_check(fv3_lagrangian_to_eulerian(at%ctx, rp, kord_tr, at%ps, at%pe, at%delp, at%pkz, at%pk, at%u, at%v, &
                                                  c_null_ptr, c_null_ptr, at%pt, at%q, at%peln, at%omga, c_null_ptr), &
                       'lagrangian_to_eulerian')
      else
        call fv3_check(fv3_lagrangian_to_eulerian(at%ctx, rp, kord_tr, at%ps, at%pe, at%delp, at%pkz, at%pk, at%u, at%v, &
                                                  at%w, at%delz, at%pt, at%q, at%peln, at%omga, at%ws), &
                       'lagrangian_to_eulerian')                                                           ! :607
      end if
    end do
  end subroutine

  !> A whole fv_dynamics call (model/fv_dynamics.F90:79-936 for the adiabatic core) on one doubly periodic tile (or this rank's block of
  !> it); pt holds T (T_v) on entry and on return.  compute_total_energy when consv_te > 0 (:345-355), T -> theta_v with the virtual
  !> effect (:296-329, :379-399), Rayleigh_Friction when tau > 0 (:372-375, :1126-1264: u2f, its halo update, the damping), the k_split
  !> loop whose last remap returns T and -- with |consv_te| > consv_min -- runs the energy fixer (fv_mapz.F90:643-772 with the
  !> reproducing sum behind fv3_ordered_sum, :793-821), cubed_to_latlon (:911).  The calls of the Python host's
  !> FvDynamics.step_from_temperature in its order.
  subroutine fv3_fv_dynamics_call(at, bdt, consv_te, tau, rf_cutoff, zvir, c2l_ord, moist_phys, radius)
    type(fv3_atmos), intent(inout) :: at
    real(c_double), intent(in) :: bdt, consv_te, tau, rf_cutoff, zvir, radius
    integer, intent(in) :: c2l_ord
    logical, intent(in) :: moist_phys
    real(c_double), parameter :: consv_min = 0.001d0, pi = 3.1415926535897931d0
    type(fv3_remap_params) :: rp
    type(c_ptr) :: qv, wp, dzp, pep, pelnp, pkp, zs0p
    real(c_double) :: zv, zsum, dtmp, ph1, ph2
    logical :: hyd, fixer
    integer(c_int) :: ihyd
    integer :: k
    hyd = at%fl%hydrostatic; ihyd = merge(1_c_int, 0_c_int, hyd)
    fixer = abs(consv_te) > consv_min
    rp%hydrostatic = ihyd; rp%adiabatic = merge(1_c_int, 0_c_int, at%fl%adiabatic); rp%nq = int(at%nq, c_int)
    rp%kord_mt = int(at%fl%kord_mt, c_int); rp%kord_wz = int(at%fl%kord_wz, c_int); rp%kord_tm = int(at%fl%kord_tm, c_int)
    rp%sphum = merge(1_c_int, 0_c_int, at%nq > 0); rp%fill = merge(1_c_int, 0_c_int, at%fl%fill)
    rp%akap = at%fl%akap; rp%ptop = at%fl%ptop; rp%rdgas = at%fl%rdgas; rp%grav = at%fl%grav
    rp%cv_air = at%fl%cp_air - at%fl%rdgas; rp%r_vir = at%fl%r_vir; rp%cp = at%fl%cp_air; rp%t_min = at%fl%t_min
    rp%last_step = 0_c_int
    if (fixer .and. .not. c_associated(at%te0)) then
      call dmalloc(at%te0, at%nCC); call dmalloc(at%te, at%nCC); call dmalloc(at%zs1, at%nCC); call dmalloc(at%zs0, at%nCC)
      call dzero(at, at%te0, at%nCC); call dzero(at, at%te, at%nCC); call dzero(at, at%zs1, at%nCC); call dzero(at, at%zs0, at%nCC)
    end if
    qv = c_null_ptr; zv = 0.d0
    if (at%nq > 0 .and. .not. at%fl%adiabatic) then
      qv = at%q; zv = zvir
    end if
    wp = at%w; dzp = at%delz; pep = c_null_ptr; pelnp = c_null_ptr
    if (hyd) then
      wp = c_null_ptr; dzp = c_null_ptr; pep = at%pe; pelnp = at%peln
    end if
    if (at%fl%use_cond .or. at%fl%moist_kappa) then    ! moist_cv of the conversions below and of compute_total_energy (fv_dynamics.F90:305-317)
      at%fl%moist%moist_kappa = merge(1_c_int, 0_c_int, at%fl%moist_kappa)
      at%fl%moist%use_cond = merge(1_c_int, 0_c_int, at%fl%use_cond)
      call fv3_check(fv3_set_moist(at%ctx, at%fl%moist, at%q_con, at%cappa), 'set_moist')
    end if
    if (consv_te > consv_min) &
      call fv3_check(fv3_compute_total_energy(at%ctx, rp, merge(1_c_int, 0_c_int, moist_phys), at%u, at%v, wp, dzp, at%pt, at%delp, &
                                              at%q, c_null_ptr, pep, pelnp, at%phis, at%te0), 'compute_total_energy')
    if (at%consv_am) call aam(at%am_teq, at%am_ps2)                      ! :358-361: teq, ps2 of the state the step starts from
    if (tau > 0.d0) then
      if (at%kmax < 0) then      ! rf(k), kmax (:1169-1182) with pfull of :254-262
        if (allocated(at%rf)) deallocate(at%rf, at%pm)
        allocate(at%rf(at%npz), at%pm(at%npz)); at%rf = 0.d0; at%kmax = 0
        do k = 1, at%npz
          ph1 = at%ak(k) + at%bk(k) * 1.d5; ph2 = at%ak(k+1) + at%bk(k+1) * 1.d5
          at%pm(k) = (ph2 - ph1) / log(ph2 / ph1)
        end do
        do k = 1, at%npz
          if (at%pm(k) < rf_cutoff) then
            at%rf(k) = abs(bdt) / (tau * 86400.d0) * sin(0.5d0 * pi * log(rf_cutoff / at%pm(k)) / log(rf_cutoff / at%fl%ptop))**2
            at%kmax = k
          else
            exit
          end if
        end do
      end if
      if (.not. hyd) call fv3_check(fv3_pt_to_theta_v(at%ctx, -1_c_int, zv, at%fl%akap, at%fl%rdgas, at%fl%grav, at%pt, at%delp, at%delz, &
                                                      qv, at%pkz), 'pt_to_theta_v')
      if (at%kmax > 0) then
        if (.not. c_associated(at%u2f)) then
          call dmalloc(at%u2f, at%nA * at%npz); call dzero(at, at%u2f, at%nA * at%npz)
        end if
        call fv3_check(fv3_rayleigh_u2f(at%ctx, int(at%kmax, c_int), ihyd, at%u, at%v, wp, at%ua, at%va, at%u2f), 'rayleigh_u2f')
        call halo(at, at%u2f, KIND_A, at%npz)                                                               ! :1207-1209
        call fv3_check(fv3_rayleigh_apply(at%ctx, int(at%kmax, c_int), 1_c_int, ihyd, at%fl%cp_air, at%fl%rdgas, at%fl%ptop, at%pm, at%rf, &
                                          at%u2f, at%pt, dzp, at%u, at%v, wp), 'rayleigh_apply')
      end if
      call fv3_check(fv3_pt_to_theta_v(at%ctx, 1_c_int, zv, at%fl%akap, at%fl%rdgas, at%fl%grav, at%pt, at%delp, dzp, qv, at%pkz), &
                     'pt_to_theta_v')
    else
      call fv3_check(fv3_pt_to_theta_v(at%ctx, ihyd, zv, at%fl%akap, at%fl%rdgas, at%fl%grav, at%pt, at%delp, dzp, qv, at%pkz), &
                     'pt_to_theta_v')
    end if
    if (fixer) then
      call fv3_fv_dynamics(at, bdt, .true., 2)
      pkp = c_null_ptr; zs0p = c_null_ptr
      if (hyd) then
        pkp = at%pk; zs0p = at%zs0
      end if
      rp%last_step = 2_c_int
      call fv3_check(fv3_energy_fixer_sums(at%ctx, rp, merge(1_c_int, 0_c_int, consv_te < 0.d0), at%u, at%v, wp, dzp, at%pt, at%delp, &
                                           at%q, pep, pelnp, at%phis, at%pkz, pkp, at%te0, at%te, at%zs1, zs0p), 'energy_fixer_sums')
      if (hyd) then
        zsum = g_sum(at%zs0)
      else
        zsum = g_sum(at%zs1)
      end if
      if (consv_te < 0.d0) then
        at%e_flux = consv_te
        dtmp = at%e_flux * (at%fl%grav * bdt * 4.d0 * pi * radius**2) / zsum
      else
        dtmp = consv_te * g_sum(at%te)
        at%e_flux = dtmp / (at%fl%grav * bdt * 4.d0 * pi * radius**2)
        dtmp = dtmp / zsum
      end if
      at%dtmp = dtmp
      call fv3_check(fv3_remap_finish(at%ctx, rp, dtmp, at%pt, at%pkz, at%q), 'remap_finish')
    else
      call fv3_fv_dynamics(at, bdt, .true.)
    end if
    if (at%consv_am) call consv_am_correct()                             ! :747-800
    if (c2l_ord == 4) then                                               ! fv_grid_utils.F90:2372-2376
      call halo(at, at%u, KIND_U, at%npz); call halo(at, at%v, KIND_V, at%npz)
    end if
    call fv3_check(fv3_c2l(at%ctx, int(c2l_ord, c_int), at%u, at%v, at%ua, at%va), 'c2l')      ! :911
  contains
    !> compute_aam (fv_dynamics.F90:1266-1314): cubed_to_latlon (mode 1, c2l_ord 2: no halo update), then aam, m_fac, ps of every column
    subroutine aam(aam_d, ps_d)
      type(c_ptr), intent(in) :: aam_d, ps_d
      call fv3_check(fv3_c2l(at%ctx, 2_c_int, at%u, at%v, at%ua, at%va), 'c2l (compute_aam)')              ! :1287
      call fv3_check(fv3_compute_aam(at%ctx, radius, at%am_omega, 1.d0 / at%fl%grav, at%fl%ptop, at%am_coslat, at%ua, at%delp, &
                                     aam_d, at%am_mfac, ps_d), 'compute_aam')
    end subroutine
    !> :747-800: te_2d = aam - teq + dt2 (ps2 + ps) zxg, the two reproducing global sums, u00, u += u00 l2c_u, v += u00 l2c_v
    subroutine consv_am_correct()
      real(c_double), allocatable, target :: te(:,:), teq(:,:), ps2(:,:), ps1(:,:), te2(:,:)
      real(c_double) :: amdt
      integer :: ng_
      call aam(at%am_aam, at%ps)
      ng_ = at%is - at%isd
      allocate(te(at%nx, at%ny), teq(at%nx, at%ny), te2(at%nx, at%ny))
      allocate(ps2(at%isd:at%ied, at%jsd:at%jed), ps1(at%isd:at%ied, at%jsd:at%jed))
      call fv3_check(fv3_memcpy_d2h(at%ctx, c_loc(te), at%am_aam, at%nCC * 8_c_size_t), 'd2h')
      call fv3_check(fv3_memcpy_d2h(at%ctx, c_loc(teq), at%am_teq, at%nCC * 8_c_size_t), 'd2h')
      call fv3_check(fv3_memcpy_d2h(at%ctx, c_loc(ps2), at%am_ps2, at%nA * 8_c_size_t), 'd2h')
      call fv3_check(fv3_memcpy_d2h(at%ctx, c_loc(ps1), at%ps, at%nA * 8_c_size_t), 'd2h')
      call fv3_check(fv3_sync(at%ctx), 'sync')
      te2 = te - teq + (0.5d0 * bdt) * (ps2(at%is:at%ie, at%js:at%je) + ps1(at%is:at%ie, at%js:at%je)) * at%am_zxg        ! :761-767
      call fv3_check(fv3_memcpy_h2d(at%ctx, at%am_aam, c_loc(te2), at%nCC * 8_c_size_t), 'h2d')
      amdt = g_sum(at%am_aam)                                                                                              ! :771
      at%u00 = -radius * amdt / g_sum(at%am_mfac)                                                                          ! :772
      call fv3_check(fv3_consv_am_apply(at%ctx, at%u00, at%am_l2c_u, at%am_l2c_v, at%u, at%v), 'consv_am_apply')           ! :784-798
    end subroutine
    function g_sum(col) result(tot)      ! g_sum(..., area, 0, reproduce = .true.) over this rank's block and the ranks
      type(c_ptr), intent(in) :: col
      real(c_double) :: tot
      real(c_double), allocatable, target :: h(:,:), vals(:)
      allocate(h(at%nx, at%ny), vals(at%nx * at%ny))
      call fv3_check(fv3_memcpy_d2h(at%ctx, c_loc(h), col, at%nCC * 8_c_size_t), 'd2h')
      call fv3_check(fv3_sync(at%ctx), 'sync')
      vals = reshape(h * at%area, [at%nx * at%ny])
      call fv3_check(fv3_ordered_sum(at%ctx, vals, int(size(vals), c_size_t), tot), 'ordered_sum')
    end function
  end subroutine

  !> inline_q (dyn_core.F90:340 / :573 / :768, sw_core.F90:1020-1043), before d_sw: the halo of q; d_sw gets zeroed flux arrays
  !> of its own (fxp, fyp) so that they hold this substep's delp fluxes afterwards.  Without inline_q: fxp, fyp = mfx, mfy
  subroutine inline_q_begin(at, fxp, fyp, skip_halo)
    type(fv3_atmos), intent(inout) :: at
    type(c_ptr), intent(out) :: fxp, fyp
    logical, intent(in), optional :: skip_halo       ! the caller exchanged q itself (the six faces of the sphere)
    logical :: do_halo
    fxp = at%mfx; fyp = at%mfy
    if (.not. (at%fl%inline_q .and. at%nq > 0)) return
    do_halo = .true.
    if (present(skip_halo)) do_halo = .not. skip_halo
    if (do_halo) call halo(at, at%q, KIND_A, at%npz * at%nq)                                  ! :341 start ... :573 complete
    call dzero(at, at%fx_s, at%nFX * at%npz); call dzero(at, at%fy_s, at%nFY * at%npz)
    fxp = at%fx_s; fyp = at%fy_s
  end subroutine

  !> ... after d_sw and before the swap of delp: q -> q_n, mfx += fx
  subroutine inline_q_end(at)
    type(fv3_atmos), intent(inout) :: at
    real(c_double) :: damp_t
    integer(c_int) :: nord_t
    if (.not. (at%fl%inline_q .and. at%nq > 0)) return
    nord_t = int(min(2, at%fl%nord), c_int)                                                   ! dyn_core.F90:679, :690
    damp_t = merge(at%fl%vtdm4, 0.d0, at%fl%do_vort_damp)                                     ! :683-687, :692
    call fv3_check(fv3_d_sw_inline_q(at%ctx, int(at%nq, c_int), int(at%fl%hord_tr, c_int), nord_t, damp_t, at%q, at%q_n, &
                                     at%delp, at%delp_n, at%fx_s, at%fy_s, at%crx, at%cry, at%xfx, at%yfx), 'd_sw_inline_q')
    call fv3_check(fv3_flux_accum(at%ctx, at%mfx, at%mfy, at%fx_s, at%fy_s), 'flux_accum')
    call swap(at%q, at%q_n)
  end subroutine

  !> the condensate loading and the moist kappa of the Riemann solvers and of the heating (fv3_set_condensate): the current q_con buffer
  !> pfull(k) of fv_dynamics.F90:254-262 (p_ref = 1e5)
  pure function host_pfull(at, k) result(pf)
    type(fv3_atmos), intent(in) :: at
    integer, intent(in) :: k
    real(c_double) :: pf, ph1, ph2
    if (allocated(at%pfull)) then
      pf = at%pfull(k)
      return
    end if
    ph1 = at%ak(k) + at%bk(k) * 1.d5; ph2 = at%ak(k+1) + at%bk(k+1) * 1.d5
    pf = (ph2 - ph1) / log(ph2 / ph1)
  end function

  !> fast_tau_w_sec > 0: rff(k) on Riem_Solver_c's first call, with ITS dt (nh_utils.F90:356-367), handed to the library once
  subroutine host_fast_tau_w(at, dt_c)
    type(fv3_atmos), intent(inout) :: at
    real(c_double), intent(in) :: dt_c
    real(c_double), parameter :: pi_8 = 3.14159265358979323846d0
    real(c_double) :: rff(at%npz), rff_temp
    integer :: k, k_rf
    if (.not. (at%fl%fast_tau_w_sec > 1.d-5) .or. at%rfw_ready) return
    k_rf = 0; rff = 1.d0
    do k = 1, at%npz
      if (host_pfull(at, k) > at%fl%rf_cutoff) exit
      k_rf = k
      rff_temp = dt_c / at%fl%fast_tau_w_sec * sin(0.5d0 * pi_8 * log(at%fl%rf_cutoff / host_pfull(at, k)) / log(at%fl%rf_cutoff / at%fl%ptop))**2
      rff(k) = 1.0d0 / (1.0d0 + rff_temp)
    end do
    call fv3_check(fv3_set_fast_tau_w(at%ctx, int(k_rf, c_int), rff), 'set_fast_tau_w')
    at%rfw_ready = .true.
  end subroutine

  !> Ray_fast (dyn_core.F90:2485-2601) when RF_fast .and. tau > 0 (:1057-1060): the profile of the first call (:2519-2545), then the kernel
  subroutine host_ray_fast(at, dt)
    type(fv3_atmos), intent(inout) :: at
    real(c_double), intent(in) :: dt
    real(c_double), parameter :: sday = 86400.d0, pi = 3.1415926535897931d0
    real(c_double) :: rf(at%npz), dp(at%npz), rffk, tau0, dm
    integer :: k, kmax, k_rf
    type(c_ptr) :: wp
    if (.not. (at%fl%RF_fast .and. at%fl%tau > 0.d0)) return
    if (.not. at%rff_ready) then
      tau0 = at%fl%tau * sday
      rf = 1.d0; kmax = 1
      do k = 1, at%npz
        dp(k) = (at%ak(k+1) - at%ak(k)) + (at%bk(k+1) - at%bk(k)) * 1.d5                  ! dp_ref, dyn_core.F90:241-244
      end do
      do k = 1, at%npz
        if (host_pfull(at, k) < at%fl%rf_cutoff) then
          rffk = abs(dt) / tau0 * sin(0.5d0 * pi * log(at%fl%rf_cutoff / host_pfull(at, k)) / log(at%fl%rf_cutoff / at%fl%ptop))**2
          kmax = k
          rf(k) = 1.d0 / (1.0d0 + rffk)
        else
          exit
        end if
      end do
      dm = 0.d0; k_rf = 0
      do k = 1, at%fl%ks
        if (host_pfull(at, k) < at%fl%rf_cutoff + min(100.d0, 10.d0 * at%fl%ptop)) then
          dm = dm + dp(k)
          k_rf = k
        else
          exit
        end if
      end do
      if (k_rf == 0) dm = 1.d0
      call fv3_check(fv3_set_ray_fast(at%ctx, int(kmax, c_int), int(k_rf, c_int), dm, rf, dp), 'set_ray_fast')
      at%rff_ready = .true.
    end if
    wp = at%w
    if (at%fl%hydrostatic) wp = c_null_ptr
    call fv3_check(fv3_ray_fast(at%ctx, at%u, at%v, wp, merge(1_c_int, 0_c_int, at%fl%hydrostatic)), 'ray_fast')
  end subroutine

  subroutine set_condensate(at)
    type(fv3_atmos), intent(in) :: at
    type(c_ptr) :: qc, cp
    if (.not. (at%fl%use_cond .or. at%fl%moist_kappa)) return
    qc = c_null_ptr; cp = c_null_ptr
    if (at%fl%use_cond) qc = at%q_con
    if (at%fl%moist_kappa) cp = at%cappa
    call fv3_check(fv3_set_condensate(at%ctx, qc, cp), 'set_condensate')
  end subroutine

  !> flagstruct%consv_am: what compute_aam and the correction read of the grid and of the diagnostics (fv_dynamics.F90:1266-1314,
  !> :761-798): coslat = cos(agrid(:,:,2)) on (isd:ied, jsd:jed); l2c_u on (isd:ied, jsd:jed+1), l2c_v on (isd:ied+1, jsd:jed) (the
  !> reference's members cover the compute domain: the caller pads with zeros); zxg on the compute domain
  subroutine fv3_host_set_consv_am(at, coslat, l2c_u, l2c_v, zxg, omega)
    type(fv3_atmos), intent(inout) :: at
    real(c_double), intent(in), target, contiguous :: coslat(:,:), l2c_u(:,:), l2c_v(:,:)
    real(c_double), intent(in) :: zxg(:,:)
    real(c_double), intent(in), optional :: omega
    if (.not. c_associated(at%am_coslat)) then
      call dmalloc(at%am_coslat, at%nA); call dmalloc(at%am_l2c_u, at%nU); call dmalloc(at%am_l2c_v, at%nV)
      call dmalloc(at%am_teq, at%nCC);   call dmalloc(at%am_aam, at%nCC);  call dmalloc(at%am_mfac, at%nCC); call dmalloc(at%am_ps2, at%nA)
      call dzero(at, at%am_teq, at%nCC); call dzero(at, at%am_aam, at%nCC); call dzero(at, at%am_mfac, at%nCC); call dzero(at, at%am_ps2, at%nA)
    end if
    if (size(coslat) /= at%nA .or. size(l2c_u) /= at%nU .or. size(l2c_v) /= at%nV .or. size(zxg, 1) /= at%nx .or. size(zxg, 2) /= at%ny) &
      error stop 'fv3_host_set_consv_am: coslat (A), l2c_u (U), l2c_v (V) with halos, zxg on the compute domain'
    call fv3_check(fv3_memcpy_h2d(at%ctx, at%am_coslat, c_loc(coslat), at%nA * 8_c_size_t), 'h2d')
    call fv3_check(fv3_memcpy_h2d(at%ctx, at%am_l2c_u, c_loc(l2c_u), at%nU * 8_c_size_t), 'h2d')
    call fv3_check(fv3_memcpy_h2d(at%ctx, at%am_l2c_v, c_loc(l2c_v), at%nV * 8_c_size_t), 'h2d')
    call fv3_check(fv3_sync(at%ctx), 'sync')
    at%am_zxg = zxg
    if (present(omega)) at%am_omega = omega
    at%consv_am = .true.
  end subroutine

  subroutine fv3_host_final(at)
    type(fv3_atmos), intent(inout) :: at
    call fv3_check(fv3_sync(at%ctx), 'fv3_sync')
    call fv3_check(fv3_destroy(at%ctx), 'fv3_destroy')     ! device arrays from fv3_malloc are the caller's: freed with the process
    at%ctx = c_null_ptr
  end subroutine

end module fv3_host_mod
