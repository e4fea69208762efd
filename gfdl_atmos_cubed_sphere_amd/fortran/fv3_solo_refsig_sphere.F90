!> Solo driver of fv_dynamics WITH THE REFERENCE'S ARGUMENT LIST (fv3_dyn_core_mod, model/fv_dynamics.F90:79-85) on the CUBED SPHERE:
!> the model's side of the call -- host arrays with the fv_arrays layout, gridstruct / flagstruct / bd / domain of the tile -- for the
!> tiles this process holds, one call per tile as the reference makes it.  One process can hold all six tiles (one GPU for the
!> sphere) or the tiles can be spread over processes (rank / nranks / face_rank: the exchange then runs between the processes).
!> usage: fv3_solo_refsig_sphere <input file> <output file>    (raw little-endian streams; the output gets ".<rank>" appended)
!>
!> input : int32  npx, npz, nq, n_split, k_split, mode (bit 0: hydrostatic; bit 3: thermostruct%use_cond = moist_kappa = .true.; bit 4:
!>                flagstruct%do_diss_est with prevent_diss_cooling off -- diss_est joins the output; bit 5: flagstruct%fill_dp; bit 6:
!>                flagstruct%consv_am -- needs have_grid (agrid); l2c_u, l2c_v, zxg of every tile follow its pe .. pkz), nord, rank, nranks, have_grid, face_rank(6)
!>         real64 bdt, ptop, d_con, d_ext, da_min, da_min_c, d4_bg, beta, consv_te, tau, zvir
!>         int8   comm_id(128) ; real64 ak(npz+1), bk(npz+1)
!>         per tile (six times): the gridstruct members in the order of fv3_grid_host, then edge_w, edge_e, edge_s, edge_n, rsina,
!>         corner_f(12), a11 .. a22 (A layout), ec1, ec2 (A x 3), en1, en2 (component last) [, grid, agrid when have_grid: then
!>         corner_f is NOT handed over and fv3_dyn_core_mod forms it from grid / agrid];
!>         then the state: u, v, w, delp, pt (TEMPERATURE), delz, phis, q, and pe, pk, peln, pkz as p_var left them [, q_con, cappa in the
!>         moist dyn_core mode: fv_dynamics forms them itself]
!> output: per tile held: u, v, w, delp, pt, delz, q, ua, va, mfx, cx [, q_con]
program fv3_solo_refsig_sphere
  use iso_c_binding
  use fv3_arrays_compat_mod
  use fv3_dyn_core_mod
  implicit none
  character(len=1024) :: fin, fout
  character(len=16) :: sfx, what
  type(group_halo_update_type) :: i_pack(13)
  real(c_double), allocatable :: pfull(:), te0(:,:), cappa(:,:,:)
  integer(c_int) :: npx, npz, nq, n_split, k_split, mode, nord, rank, nranks, have_grid, face_rank(6)
  real(c_double) :: bdt, ptop, d_con, d_ext, da_min, da_min_c, d4_bg, beta, consv_te, tau, zvir
  integer(c_signed_char) :: comm_id(128)
  real(c_double), allocatable :: ak(:), bk(:)
  type tile_state
    real(c_double), allocatable :: u(:,:,:), v(:,:,:), w(:,:,:), delp(:,:,:), pt(:,:,:), delz(:,:,:), phis(:,:), q(:,:,:,:)
    real(c_double), allocatable :: ps(:,:), pe(:,:,:), pk(:,:,:), peln(:,:,:), pkz(:,:,:), omga(:,:,:), ua(:,:,:), va(:,:,:)
    real(c_double), allocatable :: uc(:,:,:), vc(:,:,:), mfx(:,:,:), mfy(:,:,:), cx(:,:,:), cy(:,:,:), qcon(:,:,:), ze0(:,:,:)
    real(c_double), allocatable :: heat(:,:,:), diss(:,:,:), cappa(:,:,:)
  end type
  type(tile_state), target :: st(6)
  type(fv_grid_type), target :: gs(6)
  type(fv_grid_bounds_type) :: bd
  type(fv_flags_type) :: fl
  type(fv_nest_type) :: nest
  type(fv_thermo_type) :: thermo
  type(fv_diag_type) :: idiag, idg(6)
  type(domain2d) :: dom
  type(fv_atmos_type), pointer :: parent => null()
  type(inline_mp_type) :: imp
  real(c_double), allocatable :: a9(:,:,:), u9(:,:,:), v9(:,:,:), b4(:,:,:), a4(:,:,:), ec(:,:,:,:), en1(:,:,:), en2(:,:,:)
  logical :: hydrostatic, moist
  integer :: un, t, nx, isd, ied, c

  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  call get_command_argument(3, what)          ! "dyn_core": one dyn_core call (pt = theta_v) in the place of the fv_dynamics call
  open(newunit=un, file=trim(fin), access='stream', form='unformatted', status='old')
  read(un) npx, npz, nq, n_split, k_split, mode, nord, rank, nranks, have_grid, face_rank
  read(un) bdt, ptop, d_con, d_ext, da_min, da_min_c, d4_bg, beta, consv_te, tau, zvir
  read(un) comm_id
  allocate(ak(npz+1), bk(npz+1))
  read(un) ak, bk
  hydrostatic = iand(mode, 1_c_int) /= 0
  moist = iand(mode, 8_c_int) /= 0
  nx = npx - 1; isd = 1 - 3; ied = nx + 3
  bd%is = 1; bd%ie = nx; bd%js = 1; bd%je = nx; bd%isd = isd; bd%ied = ied; bd%jsd = isd; bd%jed = ied
  bd%isc = 1; bd%iec = nx; bd%jsc = 1; bd%jec = nx
  allocate(a9(isd:ied, isd:ied, 9), u9(isd:ied, isd:ied+1, 9), v9(isd:ied+1, isd:ied, 9), b4(isd:ied+1, isd:ied+1, 4))
  allocate(a4(isd:ied, isd:ied, 4), ec(isd:ied, isd:ied, 3, 2), en1(nx, npx, 3), en2(npx, nx, 3))
  do t = 1, 6
    associate (g => gs(t), s => st(t))
      allocate(g%area(isd:ied,isd:ied), g%rarea(isd:ied,isd:ied), g%dxa(isd:ied,isd:ied), g%dya(isd:ied,isd:ied), g%rdxa(isd:ied,isd:ied))
      allocate(g%rdya(isd:ied,isd:ied), g%cosa_s(isd:ied,isd:ied), g%rsin2(isd:ied,isd:ied), g%f0(isd:ied,isd:ied))
      allocate(g%dx(isd:ied,isd:ied+1), g%rdx(isd:ied,isd:ied+1), g%dyc(isd:ied,isd:ied+1), g%rdyc(isd:ied,isd:ied+1), g%cosa_v(isd:ied,isd:ied+1))
      allocate(g%sina_v(isd:ied,isd:ied+1), g%rsin_v(isd:ied,isd:ied+1), g%divg_u(isd:ied,isd:ied+1), g%del6_u(isd:ied,isd:ied+1))
      allocate(g%dy(isd:ied+1,isd:ied), g%rdy(isd:ied+1,isd:ied), g%dxc(isd:ied+1,isd:ied), g%rdxc(isd:ied+1,isd:ied), g%cosa_u(isd:ied+1,isd:ied))
      allocate(g%sina_u(isd:ied+1,isd:ied), g%rsin_u(isd:ied+1,isd:ied), g%divg_v(isd:ied+1,isd:ied), g%del6_v(isd:ied+1,isd:ied))
      allocate(g%rarea_c(isd:ied+1,isd:ied+1), g%fC(isd:ied+1,isd:ied+1), g%cosa(isd:ied+1,isd:ied+1), g%sina(isd:ied+1,isd:ied+1))
      allocate(g%sin_sg(isd:ied,isd:ied,9), g%cos_sg(isd:ied,isd:ied,9))
      allocate(g%edge_w(npx), g%edge_e(npx), g%edge_s(npx), g%edge_n(npx), g%rsina(1:npx, 1:npx))
      allocate(g%a11(0:npx, 0:npx), g%a12(0:npx, 0:npx), g%a21(0:npx, 0:npx), g%a22(0:npx, 0:npx))
      allocate(g%ec1(3, isd:ied, isd:ied), g%ec2(3, isd:ied, isd:ied), g%en1(3, 1:nx, 1:npx), g%en2(3, 1:npx, 1:nx))
      read(un) a9, u9, v9, b4, g%sin_sg, g%cos_sg
      g%area = a9(:,:,1); g%rarea = a9(:,:,2); g%dxa = a9(:,:,3); g%dya = a9(:,:,4); g%rdxa = a9(:,:,5); g%rdya = a9(:,:,6)
      g%cosa_s = a9(:,:,7); g%rsin2 = a9(:,:,8); g%f0 = a9(:,:,9)
      g%dx = u9(:,:,1); g%rdx = u9(:,:,2); g%dyc = u9(:,:,3); g%rdyc = u9(:,:,4); g%cosa_v = u9(:,:,5); g%sina_v = u9(:,:,6)
      g%rsin_v = u9(:,:,7); g%divg_u = u9(:,:,8); g%del6_u = u9(:,:,9)
      g%dy = v9(:,:,1); g%rdy = v9(:,:,2); g%dxc = v9(:,:,3); g%rdxc = v9(:,:,4); g%cosa_u = v9(:,:,5); g%sina_u = v9(:,:,6)
      g%rsin_u = v9(:,:,7); g%divg_v = v9(:,:,8); g%del6_v = v9(:,:,9)
      g%rarea_c = b4(:,:,1); g%fC = b4(:,:,2); g%cosa = b4(:,:,3); g%sina = b4(:,:,4)
      read(un) g%edge_w, g%edge_e, g%edge_s, g%edge_n, g%rsina, g%corner_f, a4, ec, en1, en2
      g%a11 = a4(0:npx, 0:npx, 1); g%a12 = a4(0:npx, 0:npx, 2); g%a21 = a4(0:npx, 0:npx, 3); g%a22 = a4(0:npx, 0:npx, 4)
      do c = 1, 3
        g%ec1(c, :, :) = ec(:, :, c, 1); g%ec2(c, :, :) = ec(:, :, c, 2)
        g%en1(c, :, :) = en1(:, :, c);   g%en2(c, :, :) = en2(:, :, c)
      end do
      if (have_grid /= 0) then
        allocate(g%grid(isd:ied+1, isd:ied+1, 2), g%agrid(isd:ied, isd:ied, 2))
        read(un) g%grid, g%agrid
        g%corner_f = -1.d0              ! as in the reference: not a member; formed from grid / agrid
      end if
      g%da_min = da_min; g%da_min_c = da_min_c; g%grid_type = 0
      allocate(s%u(isd:ied, isd:ied+1, npz), s%v(isd:ied+1, isd:ied, npz), s%w(isd:ied, isd:ied, npz))
      allocate(s%delp(isd:ied, isd:ied, npz), s%pt(isd:ied, isd:ied, npz), s%delz(nx, nx, npz), s%phis(isd:ied, isd:ied))
      allocate(s%q(isd:ied, isd:ied, npz, max(1, nq)))
      read(un) s%u, s%v, s%w, s%delp, s%pt, s%delz, s%phis
      if (nq > 0) read(un) s%q
      allocate(s%ps(isd:ied, isd:ied), s%pe(0:nx+1, npz+1, 0:nx+1), s%pk(nx, nx, npz+1), s%peln(nx, npz+1, nx), s%pkz(nx, nx, npz))
      allocate(s%omga(isd:ied, isd:ied, npz), s%ua(isd:ied, isd:ied, npz), s%va(isd:ied, isd:ied, npz))
      allocate(s%uc(isd:ied+1, isd:ied, npz), s%vc(isd:ied, isd:ied+1, npz), s%mfx(nx+1, nx, npz), s%mfy(nx, nx+1, npz))
      allocate(s%cx(nx+1, isd:ied, npz), s%cy(isd:ied, nx+1, npz), s%ze0(1,1,1))
      if (moist) then
        allocate(s%qcon(isd:ied, isd:ied, npz), s%cappa(isd:ied, isd:ied, npz)); s%qcon = 0.d0; s%cappa = 0.d0
      else
        allocate(s%qcon(1,1,1), s%cappa(1,1,1))
      end if
      allocate(s%heat(isd:ied, isd:ied, npz), s%diss(isd:ied, isd:ied, npz))
      read(un) s%pe, s%pk, s%peln, s%pkz              ! what p_var left (fv_arrays layout); zeros in a nonhydrostatic test
      if (moist .and. trim(what) == 'dyn_core') read(un) s%qcon, s%cappa    ! dyn_core is handed both (fv_dynamics forms them: moist_cv)
      if (iand(mode, 64_c_int) /= 0) then                                   ! consv_am: gridstruct%l2c_u / l2c_v, idiag%zxg (fv_arrays.F90:60, :112)
        allocate(g%l2c_u(nx, nx+1), g%l2c_v(nx+1, nx), idg(t)%zxg(nx, nx))
        read(un) g%l2c_u, g%l2c_v, idg(t)%zxg
      end if
      s%ps = 0.d0; s%omga = 0.d0; s%ua = 0.d0; s%va = 0.d0
      s%uc = 0.d0; s%vc = 0.d0; s%mfx = 0.d0; s%mfy = 0.d0; s%cx = 0.d0; s%cy = 0.d0; s%heat = 0.d0; s%diss = 0.d0
    end associate
  end do
  close(un)

  fl%grid_type = 0; fl%n_split = n_split; fl%k_split = k_split; fl%nord = nord; fl%d4_bg = d4_bg
  fl%d2_bg_k1 = 0.20d0; fl%d2_bg_k2 = 0.015d0                   ! the host modules' defaults (fv3_flags), as the Python host's DynFlags
  fl%hydrostatic = hydrostatic; fl%d_con = d_con; fl%d_ext = d_ext; fl%beta = beta; fl%a_imp = 1.d0
  fl%tau = tau; fl%moist_phys = .false.; fl%adiabatic = nq == 0 .or. zvir == 0.d0
  if (iand(mode, 16_c_int) /= 0) then
    fl%do_diss_est = .true.; fl%prevent_diss_cooling = .false.
  end if
  fl%fill_dp = iand(mode, 32_c_int) /= 0
  fl%consv_am = iand(mode, 64_c_int) /= 0
  if (moist) then         ! the field table of the test: six water species in tracers 1 .. 6 (what FMS's tracer manager would answer)
    thermo%use_cond = .true.; thermo%moist_kappa = .true.; fl%nwat = 6; fl%adiabatic = .false.
    call fv3_register_tracer_index('sphum', 1);   call fv3_register_tracer_index('liq_wat', 2)
    call fv3_register_tracer_index('rainwat', 3); call fv3_register_tracer_index('ice_wat', 4)
    call fv3_register_tracer_index('snowwat', 5); call fv3_register_tracer_index('graupel', 6)
  end if
  dom%pe = rank; dom%npes = nranks; dom%face_rank = face_rank; dom%comm_id = comm_id
  allocate(pfull(npz), te0(nx, nx), cappa(1,1,1)); pfull = 0.d0; te0 = 0.d0
  do t = 1, 6
    if (face_rank(t) /= rank) cycle
    dom%tile = t
    if (trim(what) == 'dyn_core') then
      associate (s => st(t))
        call dyn_core(int(npx), int(npx), int(npz), 3, 1, int(nq), bdt, 1, int(n_split), zvir, 287.04d0/(2.d0/7.d0), 2.d0/7.d0, s%cappa, &
                      9.80d0, hydrostatic, s%u, s%v, s%w, s%delz, s%pt, s%q, s%delp, s%pe, s%pk, s%phis, te0, s%omga, ptop, pfull, &
                      s%ua, s%va, s%uc, s%vc, s%mfx, s%mfy, s%cx, s%cy, s%pkz, s%peln, s%qcon, ak, bk, 0, gs(t), fl, nest, thermo, &
                      idiag, bd, dom, .true., i_pack, .true., s%heat, s%diss, 0.d0, te0)
      end associate
      cycle
    end if
    associate (s => st(t))
      call fv_dynamics(int(npx), int(npx), int(npz), int(nq), 3, bdt, consv_te, .false., .true., 2.d0/7.d0, 287.04d0/(2.d0/7.d0), zvir, &
                       ptop, 0, max(1, int(nq)), int(n_split), 0, s%u, s%v, s%u, s%v, s%w, s%delz, hydrostatic, s%pt, s%delp, s%q, &
                       s%ps, s%pe, s%pk, s%peln, s%pkz, s%phis, s%qcon, s%omga, s%ua, s%va, s%uc, s%vc, ak, bk, s%mfx, s%mfy, &
                       s%cx, s%cy, s%ze0, .false., gs(t), fl, nest, thermo, idg(t), bd, parent, dom, imp, s%heat, s%diss)
    end associate
  end do
  write(sfx, '(a,i0)') '.', rank
  open(newunit=un, file=trim(fout)//trim(sfx), access='stream', form='unformatted', status='replace')
  do t = 1, 6
    if (face_rank(t) /= rank) cycle
    write(un) st(t)%u, st(t)%v, st(t)%w, st(t)%delp, st(t)%pt, st(t)%delz
    if (nq > 0) write(un) st(t)%q
    write(un) st(t)%ua, st(t)%va
    write(un) st(t)%mfx, st(t)%cx
    if (moist) write(un) st(t)%qcon
    if (fl%do_diss_est) write(un) st(t)%diss
  end do
  close(un)
  call fv_dynamics_end()
  call dyn_core_end()
  write(*,'(a,i0,a,i0,a,es24.16)') 'fv3_solo_refsig_sphere: rank ', rank, ' of ', nranks, ' done, sum(delp) of its first tile = ', &
    sum(st(minloc(abs(face_rank - rank), 1))%delp(1:nx, 1:nx, :))
end program fv3_solo_refsig_sphere
