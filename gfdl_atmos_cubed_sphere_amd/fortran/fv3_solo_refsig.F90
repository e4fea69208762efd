!> Driver of the reference-signature dyn_core (fv3_dyn_core_mod) on one doubly periodic tile: the caller's side of the
!> drop-in -- host arrays with the fv_arrays layout, a gridstruct / flagstruct / bd filled the way fv_control and
!> init_grid fill them for grid_type = 4 (fv_grid_tools.F90:1202-1221, fv_grid_utils.F90:427, :656-665), `nsteps` calls
!> of dyn_core(...) with the reference's 60-odd arguments.  Same input file as fv3_solo (the tracers are ignored).
!> output: real64 u, v, w, delp, pt, delz, mfx, cx, pkz
!> With a third argument `fv_dynamics`: `nsteps` calls of fv_dynamics(...) (model/fv_dynamics.F90:79-85; pt of the file is then a
!> TEMPERATURE, the tracers are read and transported; adiabatic: zvir = 0).  output: u, v, w, delp, pt, delz, q, ua
!> Arguments 4 .. 8 `rank nranks px py idfile`: this process is PE `rank` of a px x py layout of the file's (global) domain -- it takes
!> its block (halos by periodic continuation), hands domain%pe / %npes / %layout / %comm_id (the 128 bytes of idfile) to the call and
!> writes its block to <output>.<rank>.
program fv3_solo_refsig
  use iso_c_binding
  use fv3_arrays_compat_mod
  use fv3_dyn_core_mod
  implicit none
  character(len=1024) :: fin, fout, mode
  type(fv_atmos_type), pointer :: parent_grid => null()
  type(inline_mp_type) :: inline_mp
  real(c_double), allocatable :: ps(:,:), u0(:,:,:), v0(:,:,:), ze0(:,:,:)
  logical :: whole, hyb_z
  integer(c_long_long) :: tc0, tc1, tcr
  integer(c_int) :: nx, ny, npz, nq, n_split, k_split, nsteps, last_step, ihydro
  real(c_double) :: dxc_, dyc_, f0_, bdt, ptop, d_con, d_ext, beta, consv_te, tau
  real(c_double), allocatable :: ak(:), bk(:), pfull(:)
  real(c_double), allocatable, dimension(:,:,:) :: u, v, w, delp, pt, delz, cappa, q_con, heat_source, diss_est, pe, peln, pk, &
                                                   omga, uc, vc, ua, va, mfx, mfy, cx, cy, pkz
  real(c_double), allocatable :: q(:,:,:,:), phis(:,:), ws(:,:), te0_2d(:,:)
  type(fv_grid_bounds_type) :: bd
  type(fv_grid_type), target :: gs
  type(fv_flags_type), target :: fs
  type(fv_nest_type) :: ns
  type(fv_thermo_type), target :: ts
  type(fv_diag_type) :: idiag
  type(domain2d) :: domain
  type(group_halo_update_type) :: i_pack(13)
  real(c_double), parameter :: RDGAS = 287.04d0, KAPPA = 2.d0/7.d0, GRAV = 9.80d0, CP_AIR = RDGAS/KAPPA
  integer :: un, n, isd, ied, jsd, jed
  logical :: hydrostatic, moist
  real(c_double) :: zvir_ = 0.d0
  integer :: rank, nranks, px, py, gnx, gny, i0, j0
  character(len=1024) :: arg, idfile
  character(len=16) :: sfx
  character(len=64) :: envbuf
  integer :: envstat
  logical :: lazy
  integer(c_long_long) :: rstat(4)

  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  mode = ' '
  if (command_argument_count() >= 3) call get_command_argument(3, mode)
  whole = trim(mode) == 'fv_dynamics'
  consv_te = 0.d0; tau = 0.d0          ! fv_dynamics mode: FV3_SOLO_CONSV_TE / FV3_SOLO_TAU (days) in the environment
  call get_environment_variable('FV3_SOLO_CONSV_TE', arg, status=n)
  if (n == 0) read(arg, *) consv_te
  call get_environment_variable('FV3_SOLO_TAU', arg, status=n)
  if (n == 0) read(arg, *) tau
  rank = 0; nranks = 1; px = 1; py = 1
  if (command_argument_count() >= 8) then
    call get_command_argument(4, arg); read(arg, *) rank
    call get_command_argument(5, arg); read(arg, *) nranks
    call get_command_argument(6, arg); read(arg, *) px
    call get_command_argument(7, arg); read(arg, *) py
    call get_command_argument(8, idfile)
    open(newunit=un, file=trim(idfile), access='stream', form='unformatted', status='old')
    read(un) domain%comm_id
    close(un)
    domain%pe = rank; domain%npes = nranks; domain%layout = [px, py]
  end if
  open(newunit=un, file=trim(fin), access='stream', form='unformatted', status='old')
  read(un) nx, ny, npz, nq, n_split, k_split, nsteps, last_step, ihydro
  read(un) dxc_, dyc_, f0_, bdt, ptop, d_con, d_ext, beta
  allocate(ak(npz+1), bk(npz+1), pfull(npz))
  read(un) ak, bk
  hydrostatic = iand(ihydro, 1_c_int) /= 0
  moist = iand(ihydro, 24_c_int) /= 0           ! bits 3, 4: use_cond, moist_kappa (then q_con, cappa follow the state)
  bd%is = 1; bd%ie = nx; bd%js = 1; bd%je = ny; bd%ng = 3
  bd%isd = -2; bd%ied = nx + 3; bd%jsd = -2; bd%jed = ny + 3
  bd%isc = 1; bd%iec = nx; bd%jsc = 1; bd%jec = ny
  isd = bd%isd; ied = bd%ied; jsd = bd%jsd; jed = bd%jed
  allocate(u(isd:ied, jsd:jed+1, npz), v(isd:ied+1, jsd:jed, npz), w(isd:ied, jsd:jed, npz), delp(isd:ied, jsd:jed, npz))
  allocate(pt(isd:ied, jsd:jed, npz), delz(1:nx, 1:ny, npz), phis(isd:ied, jsd:jed))
  read(un) u, v, w, delp, pt, delz, phis
  if (whole .and. nq > 0) then
    allocate(q(isd:ied, jsd:jed, npz, nq))
    read(un) q
  else
    allocate(q(isd:ied, jsd:jed, npz, 1))
    q = 0.d0
  end if
  if (moist) then
    allocate(cappa(isd:ied, jsd:jed, npz), q_con(isd:ied, jsd:jed, npz))
    read(un) q_con, cappa
  else
    allocate(cappa(isd:ied, jsd:jed, 1), q_con(isd:ied, jsd:jed, 1))
    cappa = 0.d0; q_con = 0.d0
  end if
  if (iand(ihydro, 32_c_int) /= 0) then          ! bit 5: flagstruct%consv_am -- the latitudes, l2c_u / l2c_v and idiag%zxg of the test
    if (nranks > 1) error stop 'fv3_solo_refsig: consv_am is driven on one PE here'
    allocate(gs%agrid(isd:ied, jsd:jed, 2), gs%l2c_u(1:nx, 1:ny+1), gs%l2c_v(1:nx+1, 1:ny), idiag%zxg(1:nx, 1:ny))
    gs%agrid = 0.d0
    read(un) gs%agrid(:, :, 2), gs%l2c_u, gs%l2c_v, idiag%zxg
    fs%consv_am = .true.
  end if
  close(un)
  gnx = nx; gny = ny
  if (nranks > 1) then      ! this PE's block of the global domain (with its halos: they lie inside the global arrays' own halos)
    if (mod(gnx, px) /= 0 .or. mod(gny, py) /= 0) error stop 'fv3_solo_refsig: the layout must divide the domain'
    nx = gnx / px; ny = gny / py
    i0 = mod(rank, px) * nx; j0 = (rank / px) * ny
    bd%is = i0 + 1; bd%ie = i0 + nx; bd%js = j0 + 1; bd%je = j0 + ny
    bd%isd = bd%is - 3; bd%ied = bd%ie + 3; bd%jsd = bd%js - 3; bd%jed = bd%je + 3
    bd%isc = bd%is; bd%iec = bd%ie; bd%jsc = bd%js; bd%jec = bd%je
    isd = bd%isd; ied = bd%ied; jsd = bd%jsd; jed = bd%jed
    call cut3(u, isd, ied, jsd, jed + 1); call cut3(v, isd, ied + 1, jsd, jed); call cut3(w, isd, ied, jsd, jed)
    call cut3(delp, isd, ied, jsd, jed);  call cut3(pt, isd, ied, jsd, jed);    call cut3(delz, bd%is, bd%ie, bd%js, bd%je)
    call cut2(phis, isd, ied, jsd, jed)
    call cut4(q, isd, ied, jsd, jed)
    call cut3(cappa, isd, ied, jsd, jed); call cut3(q_con, isd, ied, jsd, jed)
  end if
  allocate(ps(isd:ied, jsd:jed), u0(isd:ied, jsd:jed+1, 1), v0(isd:ied+1, jsd:jed, 1), ze0(nx, ny, 1))
  ps = 0.d0; u0 = 0.d0; v0 = 0.d0; ze0 = 0.d0
  allocate(heat_source(isd:ied, jsd:jed, npz), diss_est(isd:ied, jsd:jed, npz))
  allocate(pe(0:nx+1, npz+1, 0:ny+1), peln(nx, npz+1, ny), pk(nx, ny, npz+1), ws(nx, ny), te0_2d(nx, ny))
  allocate(omga(isd:ied, jsd:jed, npz), uc(isd:ied+1, jsd:jed, npz), vc(isd:ied, jsd:jed+1, npz))
  allocate(ua(isd:ied, jsd:jed, npz), va(isd:ied, jsd:jed, npz))
  allocate(mfx(nx+1, ny, npz), mfy(nx, ny+1, npz), cx(nx+1, jsd:jed, npz), cy(isd:ied, ny+1, npz), pkz(nx, ny, npz))
  heat_source = 0.d0; diss_est = 0.d0; pe = 0.d0; peln = 0.d0; pk = 0.d0; ws = 0.d0
  te0_2d = 0.d0; omga = 0.d0; uc = 0.d0; vc = 0.d0; ua = 0.d0; va = 0.d0; mfx = 0.d0; mfy = 0.d0; cx = 0.d0; cy = 0.d0; pkz = 0.d0
  do n = 1, npz
    pfull(n) = 0.5d0 * (ak(n) + ak(n+1) + (bk(n) + bk(n+1)) * 1.d5)
  end do

  ! ---- gridstruct of the Cartesian doubly periodic domain ----
  allocate(gs%area(isd:ied, jsd:jed), gs%rarea(isd:ied, jsd:jed), gs%dxa(isd:ied, jsd:jed), gs%dya(isd:ied, jsd:jed))
  allocate(gs%rdxa(isd:ied, jsd:jed), gs%rdya(isd:ied, jsd:jed), gs%cosa_s(isd:ied, jsd:jed), gs%rsin2(isd:ied, jsd:jed))
  allocate(gs%f0(isd:ied, jsd:jed))
  allocate(gs%dx(isd:ied, jsd:jed+1), gs%rdx(isd:ied, jsd:jed+1), gs%dyc(isd:ied, jsd:jed+1), gs%rdyc(isd:ied, jsd:jed+1))
  allocate(gs%cosa_v(isd:ied, jsd:jed+1), gs%sina_v(isd:ied, jsd:jed+1), gs%rsin_v(isd:ied, jsd:jed+1))
  allocate(gs%divg_u(isd:ied, jsd:jed+1), gs%del6_u(isd:ied, jsd:jed+1))
  allocate(gs%dy(isd:ied+1, jsd:jed), gs%rdy(isd:ied+1, jsd:jed), gs%dxc(isd:ied+1, jsd:jed), gs%rdxc(isd:ied+1, jsd:jed))
  allocate(gs%cosa_u(isd:ied+1, jsd:jed), gs%sina_u(isd:ied+1, jsd:jed), gs%rsin_u(isd:ied+1, jsd:jed))
  allocate(gs%divg_v(isd:ied+1, jsd:jed), gs%del6_v(isd:ied+1, jsd:jed))
  allocate(gs%rarea_c(isd:ied+1, jsd:jed+1), gs%fC(isd:ied+1, jsd:jed+1), gs%cosa(isd:ied+1, jsd:jed+1), gs%sina(isd:ied+1, jsd:jed+1))
  allocate(gs%sin_sg(isd:ied, jsd:jed, 9), gs%cos_sg(isd:ied, jsd:jed, 9))
  gs%area = dxc_ * dyc_;  gs%rarea = 1.d0 / (dxc_ * dyc_);  gs%rarea_c = 1.d0 / (dxc_ * dyc_)
  gs%dxa = dxc_; gs%dx = dxc_; gs%dxc = dxc_;  gs%rdxa = 1.d0 / dxc_; gs%rdx = 1.d0 / dxc_; gs%rdxc = 1.d0 / dxc_
  gs%dya = dyc_; gs%dy = dyc_; gs%dyc = dyc_;  gs%rdya = 1.d0 / dyc_; gs%rdy = 1.d0 / dyc_; gs%rdyc = 1.d0 / dyc_
  gs%cosa_s = 0.d0; gs%cosa_u = 0.d0; gs%cosa_v = 0.d0; gs%cosa = 0.d0; gs%cos_sg = 0.d0
  gs%rsin2 = 1.d0; gs%sina_u = 1.d0; gs%sina_v = 1.d0; gs%rsin_u = 1.d0; gs%rsin_v = 1.d0; gs%sina = 1.d0; gs%sin_sg = 1.d0
  gs%f0 = f0_; gs%fC = f0_
  gs%divg_u = 1.d0 * dyc_ / dxc_;  gs%del6_u = 1.d0 * dxc_ / dyc_     ! sina_v*dyc/dx, sina_v*dx/dyc
  gs%divg_v = 1.d0 * dxc_ / dyc_;  gs%del6_v = 1.d0 * dyc_ / dxc_     ! sina_u*dxc/dy, sina_u*dy/dxc
  gs%da_min = dxc_ * dyc_; gs%da_min_c = dxc_ * dyc_; gs%grid_type = 4

  ! ---- flagstruct: the namelist the other hosts of the test-suite run (dyn_core.DynFlags / fv3_flags defaults) ----
  fs%grid_type = 4; fs%n_split = n_split; fs%k_split = k_split; fs%hydrostatic = hydrostatic
  fs%d2_bg_k1 = 0.20d0; fs%d2_bg_k2 = 0.015d0; fs%a_imp = 1.d0; fs%d_con = d_con; fs%d_ext = d_ext; fs%beta = beta
  fs%prevent_diss_cooling = .true.; fs%adiabatic = .true.
  ts%use_cond = iand(ihydro, 8_c_int) /= 0; ts%moist_kappa = iand(ihydro, 16_c_int) /= 0
  ! fast_tau_w_sec / RF_fast of a test: FV3_REFSIG_FAST_TAU_W (seconds), FV3_REFSIG_RF_FAST (tau in days), FV3_REFSIG_RF_CUTOFF (Pa)
  call get_environment_variable('FV3_REFSIG_FAST_TAU_W', envbuf, status=envstat)
  if (envstat == 0 .and. len_trim(envbuf) > 0) read(envbuf, *) fs%fast_tau_w_sec
  call get_environment_variable('FV3_REFSIG_RF_FAST', envbuf, status=envstat)
  if (envstat == 0 .and. len_trim(envbuf) > 0) then
    read(envbuf, *) fs%tau
    fs%RF_fast = .true.
  end if
  call get_environment_variable('FV3_REFSIG_RF_CUTOFF', envbuf, status=envstat)
  if (envstat == 0 .and. len_trim(envbuf) > 0) read(envbuf, *) fs%rf_cutoff
  ! FV3_REFSIG_DISS_EST=1: flagstruct%do_diss_est (with prevent_diss_cooling off, as the SKEB configuration has it); diss_est joins the output
  call get_environment_variable('FV3_REFSIG_DISS_EST', envbuf, status=envstat)
  if (envstat == 0 .and. trim(envbuf) == '1') then
    fs%do_diss_est = .true.; fs%prevent_diss_cooling = .false.
  end if
  ! FV3_REFSIG_HYBRID_Z=1: fv_dynamics is called with hybrid_z = .true. (as in the reference: handed on, never read)
  call get_environment_variable('FV3_REFSIG_HYBRID_Z', envbuf, status=envstat)
  hyb_z = envstat == 0 .and. trim(envbuf) == '1'
  ! FV3_REFSIG_FILL_DP=1: flagstruct%fill_dp (mix_dp after d_sw, with the file's ak / bk as the reference thicknesses)
  call get_environment_variable('FV3_REFSIG_FILL_DP', envbuf, status=envstat)
  if (envstat == 0 .and. trim(envbuf) == '1') fs%fill_dp = .true.

  if (whole) then
    fs%c2l_ord = 4; fs%tau = tau; fs%moist_phys = .false.
    if (moist) then       ! the field table of the test: six water species in tracers 1 .. 6 (what FMS's tracer manager would answer)
      fs%nwat = 6
      call fv3_register_tracer_index('sphum', 1);   call fv3_register_tracer_index('liq_wat', 2)
      call fv3_register_tracer_index('rainwat', 3); call fv3_register_tracer_index('ice_wat', 4)
      call fv3_register_tracer_index('snowwat', 5); call fv3_register_tracer_index('graupel', 6)
      fs%adiabatic = .false.; zvir_ = 0.6077d0      ! moist_cv reads the vapour: the virtual effect is on (rvgas / rdgas - 1)
    end if
    if (hydrostatic) pkz = 1.d0     ! the state p_var would have left: here the file's pt is theta already (pkz = 1 <=> T = theta)
    do n = 1, nsteps
      call fv_dynamics(gnx + 1, gny + 1, int(npz), int(nq), 3, bdt, consv_te, .false., &
                       .false., KAPPA, CP_AIR, zvir_, ptop, 0, max(1, int(nq)), int(n_split), &
                       0, u0, v0, u, v, w, delz, hydrostatic, pt, delp, q, &
                       ps, pe, pk, peln, pkz, phis, q_con, omga, ua, va, uc, vc, &
                       ak, bk, mfx, mfy, cx, cy, ze0, hyb_z, &
                       gs, fs, ns, ts, idiag, bd, &
                       parent_grid, domain, inline_mp, heat_source, diss_est)
    end do
    call fv_dynamics_end()
    sfx = ' '
    if (nranks > 1) write(sfx, '(a,i0)') '.', rank
    open(newunit=un, file=trim(fout)//trim(sfx), access='stream', form='unformatted', status='replace')
    write(un) u, v, w, delp, pt, delz
    if (nq > 0) write(un) q
    write(un) ua
    if (moist) write(un) q_con
    if (fs%do_diss_est) write(un) diss_est
    close(un)
    write(*,'(a,es24.16)') 'fv3_solo_refsig: done, sum(delp) = ', sum(delp(1:nx, 1:ny, :))
    stop
  end if
  ! FV3_REFSIG_REGISTRY=1: the lazy host-address registry -- this driver writes none of the arrays between the calls and reads them
  ! after the last one, so every array is copied to the device once and fetched once
  call get_environment_variable('FV3_REFSIG_REGISTRY', envbuf, status=envstat)
  lazy = envstat == 0 .and. trim(envbuf) == '1'
  call fv3_dyn_core_registry(lazy)
  do n = 1, nsteps
    if (n == 2 .or. nsteps == 1) call system_clock(tc0, tcr)          ! (the first call binds the context and uploads the grid)
    call dyn_core(gnx + 1, gny + 1, int(npz), 3, 1, 0, bdt, 1, int(n_split), 0.d0, CP_AIR, KAPPA, cappa, GRAV, hydrostatic, &
                  u, v, w, delz, pt, q, delp, pe, pk, phis, ws, omga, ptop, pfull, ua, va, &
                  uc, vc, mfx, mfy, cx, cy, pkz, peln, q_con, ak, bk, &
                  0, gs, fs, ns, ts, idiag, bd, domain, &
                  n == 1, i_pack, n == nsteps, heat_source, diss_est, 0.d0, te0_2d)
  end do
  if (lazy) call fv3_host_fetch(c_null_ptr)
  call system_clock(tc1)
  write(*,'(a,es12.4,a,i0,a)') 'fv3_solo_refsig: seconds per dyn_core call (host arrays in and out) = ', &
    real(tc1 - tc0, c_double) / real(tcr, c_double) / real(max(1, nsteps - 1), c_double), ' (', max(1, nsteps - 1), ' calls timed)'
  call fv3_dyn_core_registry_stats(rstat)
  write(*,'(a,4(1x,i0))') 'fv3_solo_refsig: registry (h2d copies, h2d skipped, d2h copies, d2h deferred)', rstat
  call dyn_core_end()

  sfx = ' '
  if (nranks > 1) write(sfx, '(a,i0)') '.', rank
  open(newunit=un, file=trim(fout)//trim(sfx), access='stream', form='unformatted', status='replace')
  write(un) u, v, w, delp, pt, delz, mfx, cx, pkz
  if (moist) write(un) q_con
  if (fs%do_diss_est) write(un) diss_est
  close(un)
  write(*,'(a,es24.16)') 'fv3_solo_refsig: done, sum(delp) = ', sum(delp(1:nx, 1:ny, :))
contains
  subroutine cut3(a, i1, i2, j1, j2)
    real(c_double), allocatable, intent(inout) :: a(:,:,:)
    integer, intent(in) :: i1, i2, j1, j2
    real(c_double), allocatable :: t(:,:,:)
    if (size(a, 3) == 1 .and. size(a, 1) < i2 - i1 + 1) return
    call move_alloc(a, t)
    allocate(a(i1:i2, j1:j2, lbound(t, 3):ubound(t, 3)))
    a = t(i1:i2, j1:j2, :)
  end subroutine
  subroutine cut2(a, i1, i2, j1, j2)
    real(c_double), allocatable, intent(inout) :: a(:,:)
    integer, intent(in) :: i1, i2, j1, j2
    real(c_double), allocatable :: t(:,:)
    call move_alloc(a, t)
    allocate(a(i1:i2, j1:j2))
    a = t(i1:i2, j1:j2)
  end subroutine
  subroutine cut4(a, i1, i2, j1, j2)
    real(c_double), allocatable, intent(inout) :: a(:,:,:,:)
    integer, intent(in) :: i1, i2, j1, j2
    real(c_double), allocatable :: t(:,:,:,:)
    call move_alloc(a, t)
    allocate(a(i1:i2, j1:j2, lbound(t, 3):ubound(t, 3), lbound(t, 4):ubound(t, 4)))
    a = t(i1:i2, j1:j2, :, :)
  end subroutine
end program fv3_solo_refsig
