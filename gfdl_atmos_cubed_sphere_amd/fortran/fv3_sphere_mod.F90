!> Fortran host of the hot path on the CUBED SPHERE (grid_type = 0): dyn_core (model/dyn_core.F90:94-1393, nonhydrostatic and
!> hydrostatic branches, d_con heating, non-nested) and the k_split loop of fv_dynamics (model/fv_dynamics.F90:460-665) over the
!> faces this rank holds -- all six in one process (one GPU for the whole sphere: BASELINE configs 2, 3) or one per rank (config 5:
!> face_rank places them) -- one library context per face (fv3_create with grid_type 0, fv3_grid_upload + fv3_grid_upload_cubed),
!> every halo update of the loop through the cube-edge exchange behind the C ABI (fv3_cube_halo_start / _complete: the mosaic of
!> tools/fv_mp_mod.F90:498-546 as grouped RCCL messages; fields of one group travel in ONE message per pair of faces like the
!> reference's complete=.false./.true. grouping), mpp_get_boundary of (u, v) after the last substep (dyn_core.F90:1151-1163).
!> Same kernels, same order, same arguments as the Python host (dyn_core.py / fv_dynamics.py over cubed_dyn.MultiContext): the
!> test suite runs a Jablonowski-Williamson step through both and requires identical bits.
!> The per-face state is fv3_host_mod's fv3_atmos; gridstruct comes from the caller (the reference's init_grid in an integration,
!> the harness' arrays in fv3_solo_sphere).
module fv3_sphere_mod
  use iso_c_binding
  use fv3_mi355x_mod
  use fv3_host_mod
  implicit none
  private
  public :: fv3_sphere, fv3_sphere_init_face, fv3_sphere_comm, fv3_sphere_dyn_core, fv3_sphere_fv_dynamics, fv3_sphere_final
  public :: fv3_sphere_fv_dynamics_call
  public :: fv3_sphere_halo_a, fv3_sphere_halo_moist

  type fv3_sphere
    integer :: nf = 0                       !< faces held by this rank
    integer(c_int) :: faces(6) = 0          !< their tile numbers 0..5, ascending
    integer(c_int) :: face_rank(6) = 0      !< the rank holding each tile
    type(fv3_atmos) :: f(6)
    type(c_ptr) :: ctxs(6)
    logical :: adv_pe = .true.              !< en1 / en2 were uploaded (fv3_grid_cubed): omga gets its advective part (dyn_core.F90:1195)
    type(c_ptr) :: grp = c_null_ptr         !< the faces of this rank as one launch group (fv3_group_create), when there are several
    ! fv3_sphere_fv_dynamics_call: the energy fixer's columns (te0_2d, te_2d, zsum1, zsum0: is:ie x js:je), the areas g_sum weighs
    ! them with, the profile of the Rayleigh damping (set at the first use, like RF_initialized)
    type(c_ptr) :: te0(6) = c_null_ptr, te(6) = c_null_ptr, zs1(6) = c_null_ptr, zs0(6) = c_null_ptr
    real(c_double), allocatable :: area(:,:,:)
    real(c_double), allocatable :: rf(:), pm(:)
    integer :: kmax = -1
    real(c_double) :: e_flux = 0.d0, dtmp = 0.d0
  end type

contains

  !> one face: context (grid_type = 0), gridstruct, device arrays.  dom%grid_type must be 0 .. 2; gc: the cubed-sphere members.
  subroutine fv3_sphere_init_face(sp, slot, tile, dom, gh, gc, nq, fl, ak, bk)
    type(fv3_sphere), intent(inout) :: sp
    integer, intent(in) :: slot, tile, nq
    type(fv3_domain), intent(in) :: dom
    type(fv3_grid_host), intent(in) :: gh
    type(fv3_grid_cubed), intent(in) :: gc
    type(fv3_flags), intent(in) :: fl
    real(c_double), intent(in) :: ak(dom%npz+1), bk(dom%npz+1)
    if (dom%grid_type > 2) stop 'fv3_sphere_init_face: grid_type must be 0, 1 or 2'
    call fv3_host_init_grid(sp%f(slot), dom, gh, nq, fl, ak, bk)
    call fv3_check(fv3_grid_upload_cubed(sp%f(slot)%ctx, gc), 'fv3_grid_upload_cubed')
    sp%faces(slot) = int(tile, c_int)
    sp%ctxs(slot) = sp%f(slot)%ctx
    block      ! area of the compute domain: the weights of g_sum (fv_mapz.F90:736)
      real(c_double), pointer :: a(:,:)
      integer :: nx, ny
      nx = dom%ie - dom%is + 1; ny = dom%je - dom%js + 1
      if (.not. allocated(sp%area)) allocate(sp%area(nx, ny, 6))
      call c_f_pointer(gh%area, a, [nx + 6, ny + 6])
      sp%area(:, :, slot) = a(4:nx+3, 4:ny+3)
    end block
    sp%nf = max(sp%nf, slot)
    if (.not. c_associated(gc%en1)) sp%adv_pe = .false.
  end subroutine

  !> the communicator of the cube-edge exchange on the first face's context.  id: rank 0's fv3_comm_get_unique_id, distributed by the
  !> caller (MPI_Bcast) when there are several ranks; one rank makes its own.
  subroutine fv3_sphere_comm(sp, rank, nranks, face_rank, id)
    type(fv3_sphere), intent(inout) :: sp
    integer, intent(in) :: rank, nranks, face_rank(6)
    integer(c_signed_char), intent(in), optional :: id(128)
    integer(c_signed_char) :: myid(128)
    sp%face_rank = int(face_rank, c_int)
    if (present(id)) then
      myid = id
    else
      call fv3_check(fv3_comm_get_unique_id(myid), 'fv3_comm_get_unique_id')
    end if
    call fv3_check(fv3_comm_init(sp%ctxs(1), int(rank, c_int), int(nranks, c_int), myid), 'fv3_comm_init')
    ! every face is initialised by now: the faces of this rank issue each kernel in turn -> ONE launch for all of them
    if (sp%nf > 1 .and. .not. c_associated(sp%grp)) &
      call fv3_check(fv3_group_create(sp%ctxs, int(sp%nf, c_int), sp%grp), 'fv3_group_create')
  end subroutine

  ! ---- the cube-edge exchange of one group: up to 4 entries, each a scalar field or a vector pair of every face ----
  subroutine exchange(sp, n, kinds, sel0, sel1, nks)
    type(fv3_sphere), intent(inout) :: sp
    integer, intent(in) :: n, kinds(n), sel0(n), sel1(n), nks(n)
    type(fv3_cube_field) :: fd(6*4)
    integer :: i, e
    do i = 1, sp%nf
      do e = 1, n
        fd((i-1)*n + e)%kind = int(kinds(e), c_int)
        fd((i-1)*n + e)%f0 = field_of(sp%f(i), sel0(e))
        fd((i-1)*n + e)%f1 = c_null_ptr
        if (sel1(e) > 0) fd((i-1)*n + e)%f1 = field_of(sp%f(i), sel1(e))
        fd((i-1)*n + e)%nk = int(nks(e), c_int)
        fd((i-1)*n + e)%scalar_pair = 0_c_int
      end do
    end do
    call fv3_check(fv3_cube_halo_start(int(sp%nf, c_int), sp%ctxs, sp%faces, sp%face_rank, int(n, c_int), fd), 'fv3_cube_halo_start')
    call fv3_check(fv3_cube_halo_complete(int(sp%nf, c_int), sp%ctxs), 'fv3_cube_halo_complete')
  end subroutine

  !> halo update of one cell-centred field given per face (for callers outside: phis after fv3_host_upload, init_case's
  !> mpp_update_domains(phis))
  subroutine fv3_sphere_halo_a(sp, sel, nk)
    type(fv3_sphere), intent(inout) :: sp
    integer, intent(in) :: sel, nk
    call exchange(sp, 1, [FV3_CUBE_A + 0], [sel], [0], [nk])
  end subroutine

  !> the halo updates of q_con (pack 11) and cappa (pack 12) fv_dynamics makes in front of every dyn_core call (fv_dynamics.F90:464-465, :487-488)
  subroutine fv3_sphere_halo_moist(sp)
    type(fv3_sphere), intent(inout) :: sp
    if (sp%f(1)%fl%use_cond) call exchange(sp, 1, [FV3_CUBE_A + 0], [15], [0], [sp%f(1)%npz])
    if (sp%f(1)%fl%moist_kappa) call exchange(sp, 1, [FV3_CUBE_A + 0], [16], [0], [sp%f(1)%npz])
  end subroutine

  ! field selectors (the arrays are swapped between substeps, so the pointer is looked up at every exchange)
  function field_of(at, sel) result(p)
    type(fv3_atmos), intent(in) :: at
    integer, intent(in) :: sel
    type(c_ptr) :: p
    select case (sel)
    case (1);  p = at%u
    case (2);  p = at%v
    case (3);  p = at%w
    case (4);  p = at%delp
    case (5);  p = at%pt
    case (6);  p = at%zh
    case (7);  p = at%divgd
    case (8);  p = at%uc
    case (9);  p = at%vc
    case (10); p = at%pkc
    case (11); p = at%heat_source
    case (12); p = at%q
    case (13); p = at%dp1
    case (14); p = at%phis
    case (15); p = at%q_con
    case (16); p = at%cappa
    case default; p = c_null_ptr
    end select
  end function

  !> the acoustic substep loop on the faces of this rank (dyn_core.F90:313-1286)
  subroutine fv3_sphere_dyn_core(sp, bdt, end_step)
    type(fv3_sphere), intent(inout) :: sp
    real(c_double), intent(in) :: bdt
    logical, intent(in) :: end_step
    type(fv3_dsw_params) :: par
    real(c_double) :: dt, dt2, rdt, ptk, peln1, top
    integer :: it, n_split, npz, i, n_con, nq
    logical :: remap_step, heating, hyd
    integer(c_int) :: last_call, use_logp, ihyd
    type(c_ptr) :: dv2, fxp, fyp, qcp, qcn
    type(fv3_flags) :: fl
    integer, parameter :: A = FV3_CUBE_A, B = FV3_CUBE_B, D = FV3_CUBE_D, C = FV3_CUBE_C, DE = FV3_CUBE_DEDGE
    fl = sp%f(1)%fl
    npz = sp%f(1)%npz
    nq = sp%f(1)%nq
    hyd = fl%hydrostatic
    ihyd = merge(1_c_int, 0_c_int, hyd)
    n_split = fl%n_split
    dt = bdt / real(n_split, c_double)
    dt2 = 0.5d0 * dt
    rdt = 1.d0 / dt
    heating = fl%d_con > 1.d-5                                              ! dyn_core.F90:294
    ptk = fl%ptop ** fl%akap                                                ! :222
    peln1 = log(fl%ptop)
    use_logp = merge(1_c_int, 0_c_int, fl%use_logp)
    top = merge(peln1, ptk, fl%use_logp)
    par%dt = dt; par%hord_tr = fl%hord_tr; par%hord_mt = fl%hord_mt; par%hord_vt = fl%hord_vt
    par%hord_tm = fl%hord_tm; par%hord_dp = fl%hord_dp; par%dddmp = fl%dddmp; par%d4_bg = fl%d4_bg
    par%kgb = fl%ke_bg; par%hydrostatic = ihyd; par%use_cond = merge(1_c_int, 0_c_int, fl%use_cond)
    do i = 1, sp%nf
      associate (at => sp%f(i))
        if (heating) call dzero(at, at%heat_source, at%nA*npz)
        call diss_est_begin(at)                                                         ! do_diss_est: exists (zero) from the first call on
        call dzero(at, at%mfx, at%nFX*npz); call dzero(at, at%mfy, at%nFY*npz)          ! :289-292
        call dzero(at, at%cx, at%nCX*npz);  call dzero(at, at%cy, at%nCY*npz)
      end associate
    end do
    ! fv_dynamics.F90:467-470: delp, pt (pack 1) and u, v (pack 8) before the first substep
    call exchange(sp, 2, [A, A], [4, 5], [0, 0], [npz, npz])
    call exchange(sp, 1, [D], [1], [2], [npz])
    do it = 1, n_split
      remap_step = it == n_split
      last_call = merge(1_c_int, 0_c_int, remap_step)
      if (.not. hyd) then
        call exchange(sp, 1, [A], [3], [0], [npz])                                        ! w: :350 / :432 (pack 7)
        if (it == 1) then                                                                 ! :353-389
          do i = 1, sp%nf
            call fv3_check(fv3_zh_from_delz(sp%f(i)%ctx, sp%f(i)%zs, sp%f(i)%delz, sp%f(i)%zh), 'zh_from_delz')
          end do
          call exchange(sp, 1, [A], [6], [0], [npz + 1])                                  ! gz halo (pack 5)
        end if
      end if
      do i = 1, sp%nf
        associate (at => sp%f(i))
          if (hyd) then
            call fv3_check(fv3_c_sw(at%ctx, at%delpc, at%delp, at%ptc, at%pt, at%u, at%v, c_null_ptr, at%uc, at%vc, at%ua, at%va, &
                                    c_null_ptr, at%ut, at%vt, at%divgd, int(fl%nord, c_int), dt2, 1_c_int, 1_c_int), 'c_sw')
          else
            call fv3_check(fv3_c_sw(at%ctx, at%delpc, at%delp, at%ptc, at%pt, at%u, at%v, at%w, at%uc, at%vc, at%ua, at%va, &
                                    at%omga, at%ut, at%vt, at%divgd, int(fl%nord, c_int), dt2, 0_c_int, 1_c_int), 'c_sw')   ! :439-447
          end if
        end associate
      end do
      if (fl%nord > 0) call exchange(sp, 1, [B], [7], [0], [npz])                         ! divg_d: :451 / :577 (pack 3, CORNER)
      do i = 1, sp%nf
        associate (at => sp%f(i))
          if (hyd) then
            call fv3_check(fv3_geopk(at%ctx, fl%ptop, fl%akap, fl%cp_air, ptk, at%pe, at%peln, at%delpc, at%pkc, at%gz, &
                                     at%phis, at%ptc, at%pkz, 1_c_int), 'geopk (C grid)')
          else
            call fv3_check(fv3_update_dz_c(at%ctx, dt2, at%zs, at%ut, at%vt, at%zh, at%gz, at%ws3), 'update_dz_c')   ! :514-527
            call set_condensate(at)
            call host_fast_tau_w(at, dt2)
            call fv3_check(fv3_riem_solver_c(at%ctx, dt2, at%cn, at%phis, at%omga, at%ptc, at%delpc, at%gz, at%pkc, at%ws3), &
                           'riem_solver_c')                                                ! :531
          end if
          call fv3_check(fv3_p_grad_c(at%ctx, dt2, at%delpc, at%pkc, at%gz, at%uc, at%vc, ihyd), 'p_grad_c')     ! :562
        end associate
      end do
      call exchange(sp, 1, [C], [8], [9], [npz])                                          ! uc, vc: :565 / :578 (pack 9, CGRID_NE)
      if (fl%inline_q .and. nq > 0) call exchange(sp, 1, [A], [12], [0], [npz * nq])       ! q: :341 / :573 (pack 10)
      do i = 1, sp%nf
        associate (at => sp%f(i))
          call inline_q_begin(at, fxp, fyp, skip_halo=.true.)
          if (hyd) then
            call fv3_check(fv3_d_sw(at%ctx, par, at%vt, at%delp, at%pt, at%u, at%v, c_null_ptr, at%uc, at%vc, at%ua, at%va, at%divgd, &
                                    fxp, fyp, at%cx, at%cy, at%crx, at%cry, at%xfx, at%yfx, c_null_ptr, &
                                    at%delp_n, at%pt_n, at%u_n, at%v_n, c_null_ptr, c_null_ptr, at%heat_s, at%diss_e), 'd_sw')
          else
            qcp = c_null_ptr; qcn = c_null_ptr
            if (fl%use_cond) then
              qcp = at%q_con; qcn = at%q_con_n
            end if
            call fv3_check(fv3_d_sw(at%ctx, par, at%vt, at%delp, at%pt, at%u, at%v, at%w, at%uc, at%vc, at%ua, at%va, at%divgd, &
                                    fxp, fyp, at%cx, at%cy, at%crx, at%cry, at%xfx, at%yfx, qcp, &
                                    at%delp_n, at%pt_n, at%u_n, at%v_n, at%w_n, qcn, at%heat_s, at%diss_e), 'd_sw')  ! :762
          end if
          if (heating) call fv3_check(fv3_heat_source_accum(at%ctx, at%heat_source, at%heat_s), 'heat_source_accum')   ! :798-803
          if (fl%do_diss_est) call fv3_check(fv3_heat_source_accum(at%ctx, at%diss_est, at%diss_e), 'diss_est += diss_e')   ! :805-811
          call inline_q_end(at)
          ! (the nonhydrostatic loop too when one_grad_p follows: beta < -0.1, :1029-1030)
          if (hyd .or. (fl%beta < -0.1d0 .and. fl%d_ext > 0.d0)) &
            call fv3_check(fv3_divg2_ext(at%ctx, fl%d_ext, at%delp, at%vt, at%divg2), 'divg2_ext')                  ! :745-747, :791-848
          call swap(at%delp, at%delp_n); call swap(at%pt, at%pt_n)
          call swap(at%u, at%u_n); call swap(at%v, at%v_n)
          if (.not. hyd) call swap(at%w, at%w_n)
          if (fl%use_cond) call swap(at%q_con, at%q_con_n)
          if (fl%fill_dp) then                                                            ! :820
            if (hyd) then
              call fv3_check(fv3_mix_dp(at%ctx, 1_c_int, c_null_ptr, at%delp, at%pt), 'mix_dp')
            else
              call fv3_check(fv3_mix_dp(at%ctx, 0_c_int, at%w, at%delp, at%pt), 'mix_dp')
            end if
          end if
        end associate
      end do
      call exchange(sp, 2, [A, A], [4, 5], [0, 0], [npz, npz])                            ! delp, pt: :823-824 / :851 (pack 1)
      if (fl%use_cond) call exchange(sp, 1, [A], [15], [0], [npz])                       ! q_con: :825 / :852 (pack 11)
      if (hyd) then
        do i = 1, sp%nf
          associate (at => sp%f(i))
            dv2 = c_null_ptr
            if (fl%d_ext > 0.d0) dv2 = at%divg2
            call fv3_check(fv3_geopk(at%ctx, fl%ptop, fl%akap, fl%cp_air, ptk, at%pe, at%peln, at%delp, at%pkc, at%gz, &
                                     at%phis, at%pt, at%pkz, 0_c_int), 'geopk')            ! :905-907
            if (remap_step) call fv3_check(fv3_copy_a_to_cc(at%ctx, at%pkc, at%pk, int(npz + 1, c_int)), 'pk = pkc')   ! :1001-1010
            if (fl%beta > 0.d0) then                                                         ! :1018-1019, beta_d :398-406
              call fv3_check(fv3_grad1_p_update(at%ctx, dv2, at%u, at%v, at%pkc, at%gz, dt, ptk, merge(0.d0, fl%beta, it == 1), &
                                                at%du, at%dv), 'grad1_p_update')
            else
              call fv3_check(fv3_one_grad_p(at%ctx, at%u, at%v, at%pkc, at%gz, dv2, dt, ptk), 'one_grad_p')      ! :1021
            end if
            call host_ray_fast(at, dt)                                                      ! :1057-1060
          end associate
        end do
      else
        do i = 1, sp%nf
          associate (at => sp%f(i))
            if (fl%use_cond) call set_condensate(at)           ! the buffer d_sw just wrote (its halo updated above)
            call fv3_check(fv3_update_dz_d(at%ctx, int(fl%hord_tm, c_int), at%zs, at%zh, at%zh_n, at%crx, at%cry, at%xfx, &
                                           at%yfx, at%ws, rdt), 'update_dz_d')              ! :911
            call swap(at%zh, at%zh_n)
            call fv3_check(fv3_riem_solver3(at%ctx, dt, at%cn, at%zs, at%w, at%delz, at%pt, at%delp, at%zh, at%pe, at%pkc, &
                                            at%pk3, at%pk, at%peln, at%ws, use_logp, last_call, &
                                            merge(1_c_int, 0_c_int, fl%beta < -0.1d0)), 'riem_solver3')  ! :932, fp_out :939
          end associate
        end do
        call exchange(sp, 2, [A, A], [6, 10], [0, 0], [npz + 1, npz + 1])                 ! zh, pkc: :944-950 (packs 4, 5)
        do i = 1, sp%nf
          associate (at => sp%f(i))
            if (remap_step) call fv3_check(fv3_pe_halo(at%ctx, fl%ptop, at%pe, at%delp), 'pe_halo')            ! :952-953
            call fv3_check(fv3_pk3_halo(at%ctx, fl%ptop, fl%akap, at%pk3, at%delp, use_logp), 'pk3_halo')      ! :955-959
            if (fl%beta > 0.d0) then                                                         ! :1027-1028
              call fv3_check(fv3_split_p_grad(at%ctx, at%u, at%v, at%pkc, at%zh, fl%grav, at%delp, at%pk3, &
                                              merge(0.d0, fl%beta, it == 1), dt, top, at%du, at%dv), 'split_p_grad')
            else if (fl%beta < -0.1d0) then   ! :1029-1030: pkc is the full pressure, the layer weights a2b_ord4 of delp
              dv2 = c_null_ptr
              if (fl%d_ext > 0.d0) dv2 = at%divg2
              call fv3_check(fv3_one_grad_p_nh(at%ctx, at%u, at%v, at%pkc, at%zh, dv2, at%delp, dt, fl%ptop, fl%grav), 'one_grad_p (nh)')
            else
              call fv3_check(fv3_nh_p_grad(at%ctx, at%u, at%v, at%pkc, at%zh, fl%grav, at%delp, at%pk3, dt, top), 'nh_p_grad')  ! :1032
            end if
            call host_ray_fast(at, dt)                                                      ! :1057-1060
          end associate
        end do
      end if
      if (it /= n_split) then
        call exchange(sp, 1, [D], [1], [2], [npz])                                        ! u, v: :1168-1169 (pack 8)
      else
        call exchange(sp, 1, [DE], [1], [2], [npz])                                       ! mpp_get_boundary: :1151-1163
        if (.not. hyd .and. fl%use_old_omega .and. end_step) then
          do i = 1, sp%nf
            associate (at => sp%f(i))
              ! :1182-1191: omga = (pe - pem) * rdt, pem from the delp this substep started with (= delp_n after the swap)
              call fv3_check(fv3_omga_update(at%ctx, rdt, fl%ptop, at%pe, at%delp_n, at%omga), 'omga_update')
              if (sp%adv_pe) call fv3_check(fv3_adv_pe(at%ctx, fl%ptop, at%ua, at%va, at%delp_n, at%omga), 'adv_pe')   ! :1195
            end associate
          end do
        end if
      end if
    end do
    ! dissipative heating (:296-308, :1300-1355)
    n_con = host_n_con(fl, npz)
    if (n_con /= 0 .and. heating) then
      call exchange(sp, 1, [A], [11], [0], [npz])                                         ! del2_cubed's mpp_update_domains, :2399
      do i = 1, sp%nf
        associate (at => sp%f(i))
          call fv3_check(fv3_del2_cubed(at%ctx, at%heat_source, int(npz, c_int), 0.20d0 * at%da_min, &
                                        int(min(3, fl%nord + 1), c_int)), 'del2_cubed')    ! :1301-1303
          if (hyd) then
            call fv3_check(fv3_apply_heat_source(at%ctx, int(n_con, c_int), 1_c_int, bdt, fl%delt_max, fl%cp_air, &
                                                 fl%cp_air - fl%rdgas, fl%rdgas, fl%grav, at%pt, at%heat_source, at%delp, &
                                                 c_null_ptr, at%pkz), 'apply_heat_source')
          else
            call fv3_check(fv3_apply_heat_source(at%ctx, int(n_con, c_int), 0_c_int, bdt, fl%delt_max, fl%cp_air, &
                                                 fl%cp_air - fl%rdgas, fl%rdgas, fl%grav, at%pt, at%heat_source, at%delp, &
                                                 at%delz, at%pkz), 'apply_heat_source')
          end if
        end associate
      end do
    end if
  end subroutine

  !> tracer_2d (fv_tracer2d.F90:297-557) on the faces of this rank: the Courant maximum reduced over the faces (and, with several
  !> ranks, over the ranks: fv3_allreduce_max = mp_reduce_max, :405)
  subroutine sphere_tracer_2d(sp, nranks)
    type(fv3_sphere), intent(inout) :: sp
    integer, intent(in) :: nranks
    real(c_double), allocatable :: cmax(:), cm(:), frac(:)
    integer(c_int), allocatable :: ksplt(:)
    real(c_double) :: c_global
    integer :: nsplt, it, npz, k, i, nq
    type(fv3_flags) :: fl
    fl = sp%f(1)%fl
    npz = sp%f(1)%npz; nq = sp%f(1)%nq
    allocate(cmax(npz), cm(npz), frac(npz), ksplt(npz))
    cmax = 0.d0
    do i = 1, sp%nf
      associate (at => sp%f(i))
        call fv3_check(fv3_tracer_2d_prep(at%ctx, int(fl%q_split, c_int), at%cx, at%cy, at%xfx, at%yfx, cm), 'tracer_2d_prep')   ! :362-400
      end associate
      if (i == 1) then
        cmax = cm
      else
        cmax = max(cmax, cm)
      end if
    end do
    if (fl%q_split == 0) then
      if (nranks > 1) call fv3_check(fv3_allreduce_max(sp%ctxs(1), cmax, int(npz, c_int)), 'fv3_allreduce_max')   ! :405
      if (npz /= 1) then                                                                   ! :407-412
        c_global = maxval(cmax)
      else
        c_global = cmax(1)
      end if
      nsplt = int(1.d0 + c_global)
    else
      nsplt = fl%q_split
    end if
    if (nsplt /= 1) then                                                                   ! :421-456
      do k = 1, npz
        ksplt(k) = int(1.d0 + cmax(k), c_int)
        frac(k) = 1.d0 / real(ksplt(k), c_double)
      end do
      do i = 1, sp%nf
        associate (at => sp%f(i))
          call fv3_check(fv3_tracer_2d_scale(at%ctx, frac, at%cx, at%xfx, at%mfx, at%cy, at%yfx, at%mfy), 'tracer_2d_scale')
        end associate
      end do
    else
      ksplt = 1
    end if
    if (fl%trdm2 > 1.d-4) call exchange(sp, 1, [FV3_CUBE_A + 0], [13], [0], [npz])         ! dp1_pack, :466
    do it = 1, nsplt                                                                        ! :471-541
      call exchange(sp, 1, [FV3_CUBE_A + 0], [12], [0], [npz * nq])                         ! q_pack, :474 / :536
      do i = 1, sp%nf
        associate (at => sp%f(i))
          call fv3_check(fv3_tracer_2d_step(at%ctx, int(it, c_int), int(nsplt, c_int), ksplt, int(nq, c_int), &
                                            int(fl%hord_tr, c_int), int(fl%nord_tr, c_int), fl%trdm2, at%q, at%q_n, &
                                            at%dp1, at%dp1_n, at%mfx, at%mfy, at%cx, at%cy, at%xfx, at%yfx), 'tracer_2d_step')
          call swap(at%q, at%q_n)
          if (it /= nsplt) call swap(at%dp1, at%dp1_n)
        end associate
      end do
    end do
  end subroutine

  !> one dt_atmos: the k_split loop of fv_dynamics (fv_dynamics.F90:460-665) on the faces of this rank.  pt holds theta_v;
  !> last_step makes the final remap return T (fv_mapz.F90:793-821).
  subroutine fv3_sphere_fv_dynamics(sp, bdt, last_step, nranks, last_code)
    type(fv3_sphere), intent(inout) :: sp
    real(c_double), intent(in) :: bdt
    logical, intent(in) :: last_step
    integer, intent(in) :: nranks
    integer, intent(in), optional :: last_code      !< what the last remap gets as last_step (2: the energy fixer follows, T_v stays)
    type(fv3_remap_params) :: rp
    integer(c_int), allocatable :: kord_tr(:)
    real(c_double) :: mdt
    integer :: n_map, i, nq
    type(fv3_flags) :: fl
    fl = sp%f(1)%fl
    nq = sp%f(1)%nq
    mdt = bdt / real(fl%k_split, c_double)
    allocate(kord_tr(max(1, nq))); kord_tr = int(fl%kord_tr, c_int)
    rp%hydrostatic = merge(1_c_int, 0_c_int, fl%hydrostatic); rp%adiabatic = merge(1_c_int, 0_c_int, fl%adiabatic); rp%nq = int(nq, c_int)
    rp%kord_mt = int(fl%kord_mt, c_int); rp%kord_wz = int(fl%kord_wz, c_int); rp%kord_tm = int(fl%kord_tm, c_int)
    rp%sphum = merge(1_c_int, 0_c_int, nq > 0); rp%fill = merge(1_c_int, 0_c_int, fl%fill)
    rp%akap = fl%akap; rp%ptop = fl%ptop; rp%rdgas = fl%rdgas; rp%grav = fl%grav
    rp%cv_air = fl%cp_air - fl%rdgas; rp%r_vir = fl%r_vir; rp%cp = fl%cp_air; rp%t_min = fl%t_min
    if (fl%do_diss_est) then      ! dyn_core zeroes diss_est on init_step = (n_map == 1): once per fv_dynamics call (:497, dyn_core.F90:285)
      do i = 1, sp%nf
        call diss_est_begin(sp%f(i))
        call dzero(sp%f(i), sp%f(i)%diss_est, sp%f(i)%nA * int(sp%f(i)%npz, c_size_t))
      end do
    end if
    do n_map = 1, fl%k_split
      do i = 1, sp%nf
        associate (at => sp%f(i))
          call fv3_check(fv3_memcpy_d2d(at%ctx, at%dp1, at%delp, at%nA * at%npz * 8_c_size_t), 'dp1 = delp')     ! :475-481
        end associate
      end do
      call fv3_sphere_halo_moist(sp)                                                                              ! q_con, cappa: :464-465 / :487-488
      call fv3_sphere_dyn_core(sp, mdt, n_map == fl%k_split)                                                      ! :493
      if (nq > 0 .and. .not. fl%inline_q) call sphere_tracer_2d(sp, nranks)                                                                ! :500-533
      rp%last_step = merge(1_c_int, 0_c_int, last_step .and. n_map == fl%k_split)
      if (present(last_code) .and. last_step .and. n_map == fl%k_split) rp%last_step = int(last_code, c_int)
      do i = 1, sp%nf
        associate (at => sp%f(i))
          if (fl%remap_te) call fv3_check(fv3_set_remap_te(at%ctx, 1_c_int, at%phis, at%dp1), 'set_remap_te')   ! te = dp1, :612
          if (fl%use_cond .or. fl%moist_kappa) call sphere_set_moist(at)      ! q_con is a ping-pong pair: the current buffer
          if (fl%hydrostatic) then
            call fv3_check(fv3_lagrangian_to_eulerian(at%ctx, rp, kord_tr, at%ps, at%pe, at%delp, at%pkz, at%pk, at%u, at%v, &
                                                      c_null_ptr, c_null_ptr, at%pt, at%q, at%peln, at%omga, c_null_ptr), &
                           'lagrangian_to_eulerian')
          else
            call fv3_check(fv3_lagrangian_to_eulerian(at%ctx, rp, kord_tr, at%ps, at%pe, at%delp, at%pkz, at%pk, at%u, at%v, &
                                                      at%w, at%delz, at%pt, at%q, at%peln, at%omga, at%ws), &
                           'lagrangian_to_eulerian')                                                               ! :607
          end if
        end associate
      end do
    end do
  end subroutine

  !> moist_cv's parameters and the current q_con / cappa buffers of a face (fv3_set_moist)
  subroutine sphere_set_moist(at)
    type(fv3_atmos), intent(inout) :: at
    at%fl%moist%moist_kappa = merge(1_c_int, 0_c_int, at%fl%moist_kappa)
    at%fl%moist%use_cond = merge(1_c_int, 0_c_int, at%fl%use_cond)
    call fv3_check(fv3_set_moist(at%ctx, at%fl%moist, at%q_con, at%cappa), 'set_moist')
  end subroutine

  !> A whole fv_dynamics call (model/fv_dynamics.F90:79-936 for the adiabatic core) on the faces of this rank; pt holds T (T_v) on
  !> entry and on return.  In the reference's order: compute_total_energy when consv_te > 0 (:345-355), T -> theta_v with the
  !> virtual effect zvir q(sphum) (:296-329, :379-399), Rayleigh_Super when tau > 0 (:362-371, :953-1124: the cubed sphere takes
  !> Rayleigh_Super), the k_split loop (:460-665) whose last remap returns T and -- with |consv_te| > consv_min -- runs the energy
  !> fixer (fv_mapz.F90:643-772: the column sums on the device, g_sum as the reproducing sum behind fv3_ordered_sum over the faces
  !> and ranks, dtmp applied by fv3_remap_finish :793-821), cubed_to_latlon (:911).  The same calls in the same order as the Python
  !> host (fv_dynamics.py FvDynamics.step_from_temperature): the test suite requires identical bits.
  subroutine fv3_sphere_fv_dynamics_call(sp, bdt, nranks, consv_te, tau, rf_cutoff, zvir, c2l_ord, moist_phys, radius)
    type(fv3_sphere), intent(inout) :: sp
    real(c_double), intent(in) :: bdt, consv_te, tau, rf_cutoff, zvir, radius
    integer, intent(in) :: nranks, c2l_ord
    logical, intent(in) :: moist_phys
    real(c_double), parameter :: consv_min = 0.001d0, pi = 3.1415926535897931d0       ! fv_mapz.F90:45; constants_mod
    type(fv3_flags) :: fl
    type(fv3_remap_params) :: rp
    type(c_ptr) :: qv, wp, dzp, pep, pelnp, pkp, zs0p
    integer :: i, nq, npz, k, mode
    integer(c_int) :: ihyd
    logical :: fixer, hyd
    real(c_double) :: zv, zsum, tesum, dtmp
    fl = sp%f(1)%fl; nq = sp%f(1)%nq; npz = sp%f(1)%npz; hyd = fl%hydrostatic
    ihyd = merge(1_c_int, 0_c_int, hyd)
    call sphere_remap_params(sp, rp)
    fixer = abs(consv_te) > consv_min
    if (fixer .and. .not. c_associated(sp%te0(1))) then
      do i = 1, sp%nf
        call dmalloc(sp%te0(i), sp%f(i)%nCC); call dmalloc(sp%te(i), sp%f(i)%nCC)
        call dmalloc(sp%zs1(i), sp%f(i)%nCC); call dmalloc(sp%zs0(i), sp%f(i)%nCC)
        call dzero(sp%f(i), sp%te0(i), sp%f(i)%nCC); call dzero(sp%f(i), sp%te(i), sp%f(i)%nCC)
        call dzero(sp%f(i), sp%zs1(i), sp%f(i)%nCC); call dzero(sp%f(i), sp%zs0(i), sp%f(i)%nCC)
      end do
    end if
    do i = 1, sp%nf
      associate (at => sp%f(i))
        if (fl%use_cond .or. fl%moist_kappa) call sphere_set_moist(at)   ! moist_cv of the conversions and of compute_total_energy (:305-317)
        qv = c_null_ptr; zv = 0.d0
        if (nq > 0 .and. .not. fl%adiabatic) then
          qv = at%q; zv = zvir
        end if
        wp = at%w; dzp = at%delz; pep = c_null_ptr; pelnp = c_null_ptr
        if (hyd) then
          wp = c_null_ptr; dzp = c_null_ptr; pep = at%pe; pelnp = at%peln
        end if
        if (consv_te > consv_min) &                                                            ! :345-355 -> te0_2d
          call fv3_check(fv3_compute_total_energy(at%ctx, rp, merge(1_c_int, 0_c_int, moist_phys), at%u, at%v, wp, dzp, at%pt, &
                                                  at%delp, at%q, c_null_ptr, pep, pelnp, at%phis, sp%te0(i)), 'compute_total_energy')
      end associate
    end do
    if (sp%f(1)%consv_am) call aam(.true.)                    ! :358-361: teq, ps2 of the state the step starts from
    if (tau > 0.d0) then
      if (sp%kmax < 0) call rayleigh_profile(sp, abs(bdt), tau, rf_cutoff)
      if (.not. hyd) call to_theta(-1)                      ! pkz from T and delz before the damping (:323-326)
      if (sp%kmax > 0) then
        do i = 1, sp%nf
          associate (at => sp%f(i))
            wp = at%w
            if (hyd) wp = c_null_ptr
            call fv3_check(fv3_c2l(at%ctx, 2_c_int, at%u, at%v, at%ua, at%va), 'c2l')                          ! :1040-1042
            call fv3_check(fv3_rayleigh_super(at%ctx, int(sp%kmax, c_int), merge(0_c_int, 1_c_int, fl%is_ideal_case), ihyd, &
                                              fl%cp_air, fl%rdgas, fl%ptop, sp%pm, sp%rf, at%ua, at%va, at%pt, at%u, at%v, wp, &
                                              c_null_ptr, c_null_ptr), 'rayleigh_super')
          end associate
        end do
      end if
      call to_theta(1)                                      ! :389-397 with that pkz
    else
      mode = 0
      if (hyd) mode = 1
      call to_theta(mode)
    end if
    if (fixer) then
      call fv3_sphere_fv_dynamics(sp, bdt, .true., nranks, 2)       ! the last remap leaves T_v for the fixer (last_step = 2)
      do i = 1, sp%nf
        associate (at => sp%f(i))
          wp = at%w; dzp = at%delz; pep = c_null_ptr; pelnp = c_null_ptr; pkp = c_null_ptr; zs0p = c_null_ptr
          if (hyd) then
            wp = c_null_ptr; dzp = c_null_ptr; pep = at%pe; pelnp = at%peln; pkp = at%pk; zs0p = sp%zs0(i)
          end if
          rp%last_step = 2_c_int
          call fv3_check(fv3_energy_fixer_sums(at%ctx, rp, merge(1_c_int, 0_c_int, consv_te < 0.d0), at%u, at%v, wp, dzp, at%pt, &
                                               at%delp, at%q, pep, pelnp, at%phis, at%pkz, pkp, sp%te0(i), sp%te(i), sp%zs1(i), &
                                               zs0p), 'energy_fixer_sums')
        end associate
      end do
      if (hyd) then
        zsum = g_sum(sp%zs0)
      else
        zsum = g_sum(sp%zs1)
      end if
      if (consv_te < 0.d0) then                                                                ! :745-771: a prescribed flux
        sp%e_flux = consv_te
        dtmp = sp%e_flux * (fl%grav * bdt * 4.d0 * pi * radius**2) / zsum
      else
        tesum = g_sum(sp%te)
        dtmp = consv_te * tesum
        sp%e_flux = dtmp / (fl%grav * bdt * 4.d0 * pi * radius**2)
        dtmp = dtmp / zsum
      end if
      sp%dtmp = dtmp
      do i = 1, sp%nf
        call fv3_check(fv3_remap_finish(sp%f(i)%ctx, rp, dtmp, sp%f(i)%pt, sp%f(i)%pkz, sp%f(i)%q), 'remap_finish')
      end do
    else
      call fv3_sphere_fv_dynamics(sp, bdt, .true., nranks)
    end if
    if (sp%f(1)%consv_am) call consv_am_correct()            ! :747-800
    if (c2l_ord == 4) call exchange(sp, 1, [FV3_CUBE_D + 0], [1], [2], [npz])                  ! fv_grid_utils.F90:2372-2376
    do i = 1, sp%nf
      call fv3_check(fv3_c2l(sp%f(i)%ctx, int(c2l_ord, c_int), sp%f(i)%u, sp%f(i)%v, sp%f(i)%ua, sp%f(i)%va), 'c2l')   ! :911
    end do

  contains

    !> compute_aam (fv_dynamics.F90:1266-1314) of every face: cubed_to_latlon (mode 1, c2l_ord 2: no halo update), then aam, m_fac, ps
    !> of every column; first: into teq / ps2 (the state the step starts from), else into the work array / ps
    subroutine aam(first)
      logical, intent(in) :: first
      integer :: ii
      type(c_ptr) :: aam_d, ps_d
      do ii = 1, sp%nf
        associate (at => sp%f(ii))
          aam_d = at%am_aam; ps_d = at%ps
          if (first) then
            aam_d = at%am_teq; ps_d = at%am_ps2
          end if
          call fv3_check(fv3_c2l(at%ctx, 2_c_int, at%u, at%v, at%ua, at%va), 'c2l (compute_aam)')          ! :1287
          call fv3_check(fv3_compute_aam(at%ctx, radius, at%am_omega, 1.d0 / fl%grav, fl%ptop, at%am_coslat, at%ua, at%delp, &
                                         aam_d, at%am_mfac, ps_d), 'compute_aam')
        end associate
      end do
    end subroutine

    !> :747-800: te_2d = aam - teq + dt2 (ps2 + ps) zxg on every face, the two reproducing global sums over the faces (and ranks), u00,
    !> u += u00 l2c_u, v += u00 l2c_v
    subroutine consv_am_correct()
      real(c_double), allocatable, target :: te(:,:), teq(:,:), ps2(:,:), ps1(:,:), te2(:,:)
      type(c_ptr) :: cols(6)
      real(c_double) :: amdt, u00
      integer :: ii
      call aam(.false.)
      cols = c_null_ptr
      do ii = 1, sp%nf
        associate (at => sp%f(ii))
          allocate(te(at%nx, at%ny), teq(at%nx, at%ny), te2(at%nx, at%ny))
          allocate(ps2(at%isd:at%ied, at%jsd:at%jed), ps1(at%isd:at%ied, at%jsd:at%jed))
          call fv3_check(fv3_memcpy_d2h(at%ctx, c_loc(te), at%am_aam, at%nCC * 8_c_size_t), 'd2h')
          call fv3_check(fv3_memcpy_d2h(at%ctx, c_loc(teq), at%am_teq, at%nCC * 8_c_size_t), 'd2h')
          call fv3_check(fv3_memcpy_d2h(at%ctx, c_loc(ps2), at%am_ps2, at%nA * 8_c_size_t), 'd2h')
          call fv3_check(fv3_memcpy_d2h(at%ctx, c_loc(ps1), at%ps, at%nA * 8_c_size_t), 'd2h')
          call fv3_check(fv3_sync(at%ctx), 'sync')
          te2 = te - teq + (0.5d0 * bdt) * (ps2(at%is:at%ie, at%js:at%je) + ps1(at%is:at%ie, at%js:at%je)) * at%am_zxg    ! :761-767
          call fv3_check(fv3_memcpy_h2d(at%ctx, at%am_aam, c_loc(te2), at%nCC * 8_c_size_t), 'h2d')
          call fv3_check(fv3_sync(at%ctx), 'sync')
          deallocate(te, teq, te2, ps2, ps1)
          cols(ii) = at%am_aam
        end associate
      end do
      amdt = g_sum(cols)                                                                                                   ! :771
      do ii = 1, sp%nf
        cols(ii) = sp%f(ii)%am_mfac
      end do
      u00 = -radius * amdt / g_sum(cols)                                                                                   ! :772
      do ii = 1, sp%nf
        sp%f(ii)%u00 = u00
        call fv3_check(fv3_consv_am_apply(sp%f(ii)%ctx, u00, sp%f(ii)%am_l2c_u, sp%f(ii)%am_l2c_v, sp%f(ii)%u, sp%f(ii)%v), &
                       'consv_am_apply')                                                                                   ! :784-798
      end do
    end subroutine

    subroutine to_theta(m)
      integer, intent(in) :: m
      integer :: ii
      type(c_ptr) :: q1, dz
      real(c_double) :: z1
      do ii = 1, sp%nf
        q1 = c_null_ptr; z1 = 0.d0
        if (nq > 0 .and. .not. fl%adiabatic) then
          q1 = sp%f(ii)%q; z1 = zvir
        end if
        dz = sp%f(ii)%delz
        if (hyd) dz = c_null_ptr
        call fv3_check(fv3_pt_to_theta_v(sp%f(ii)%ctx, int(m, c_int), z1, fl%akap, fl%rdgas, fl%grav, sp%f(ii)%pt, sp%f(ii)%delp, &
                                         dz, q1, sp%f(ii)%pkz), 'pt_to_theta_v')
      end do
    end subroutine

    !> g_sum(domain, p, ..., area, 0, reproduce = .true.): sum(p * area) over the faces of this rank and over the ranks, exact
    function g_sum(cols) result(tot)
      type(c_ptr), intent(in) :: cols(6)
      real(c_double) :: tot
      real(c_double), allocatable, target :: h(:,:), vals(:)
      integer :: ii, nx, ny
      nx = sp%f(1)%nx; ny = sp%f(1)%ny
      allocate(h(nx, ny), vals(nx * ny * sp%nf))
      do ii = 1, sp%nf
        call fv3_check(fv3_memcpy_d2h(sp%f(ii)%ctx, c_loc(h), cols(ii), sp%f(ii)%nCC * 8_c_size_t), 'd2h')
        call fv3_check(fv3_sync(sp%f(ii)%ctx), 'sync')
        vals((ii-1)*nx*ny + 1 : ii*nx*ny) = reshape(h * sp%area(:, :, ii), [nx * ny])
      end do
      call fv3_check(fv3_ordered_sum(sp%ctxs(1), vals, int(size(vals), c_size_t), tot), 'ordered_sum')
    end function
  end subroutine

  !> rf(k), kmax of Rayleigh_Super / Rayleigh_Friction (fv_dynamics.F90:1016-1036, :1169-1182) with pfull of :254-262 (p_ref = 1e5)
  subroutine rayleigh_profile(sp, dt, tau, rf_cutoff)
    type(fv3_sphere), intent(inout) :: sp
    real(c_double), intent(in) :: dt, tau, rf_cutoff
    real(c_double), parameter :: pi = 3.1415926535897931d0
    real(c_double) :: ph1, ph2, pm
    integer :: k, npz
    npz = sp%f(1)%npz
    if (allocated(sp%rf)) deallocate(sp%rf, sp%pm)
    allocate(sp%rf(npz), sp%pm(npz)); sp%rf = 0.d0; sp%kmax = 0
    do k = 1, npz
      ph1 = sp%f(1)%ak(k) + sp%f(1)%bk(k) * 1.d5; ph2 = sp%f(1)%ak(k+1) + sp%f(1)%bk(k+1) * 1.d5
      sp%pm(k) = (ph2 - ph1) / log(ph2 / ph1)
    end do
    do k = 1, npz
      pm = sp%pm(k)
      if (pm < rf_cutoff) then
        sp%rf(k) = dt / (tau * 86400.d0) * sin(0.5d0 * pi * log(rf_cutoff / pm) / log(rf_cutoff / sp%f(1)%fl%ptop))**2
        sp%kmax = k
      else
        exit
      end if
    end do
  end subroutine

  subroutine sphere_remap_params(sp, rp)
    type(fv3_sphere), intent(in) :: sp
    type(fv3_remap_params), intent(out) :: rp
    type(fv3_flags) :: fl
    integer :: nq
    fl = sp%f(1)%fl; nq = sp%f(1)%nq
    rp%hydrostatic = merge(1_c_int, 0_c_int, fl%hydrostatic); rp%adiabatic = merge(1_c_int, 0_c_int, fl%adiabatic); rp%nq = int(nq, c_int)
    rp%kord_mt = int(fl%kord_mt, c_int); rp%kord_wz = int(fl%kord_wz, c_int); rp%kord_tm = int(fl%kord_tm, c_int)
    rp%sphum = merge(1_c_int, 0_c_int, nq > 0); rp%fill = merge(1_c_int, 0_c_int, fl%fill)
    rp%akap = fl%akap; rp%ptop = fl%ptop; rp%rdgas = fl%rdgas; rp%grav = fl%grav
    rp%cv_air = fl%cp_air - fl%rdgas; rp%r_vir = fl%r_vir; rp%cp = fl%cp_air; rp%t_min = fl%t_min
    rp%last_step = 0_c_int
  end subroutine

  subroutine fv3_sphere_final(sp)
    type(fv3_sphere), intent(inout) :: sp
    integer :: i
    if (c_associated(sp%grp)) then
      call fv3_check(fv3_group_destroy(sp%grp), 'fv3_group_destroy')
      sp%grp = c_null_ptr
    end if
    do i = 1, sp%nf
      call fv3_host_final(sp%f(i))
    end do
    sp%nf = 0
  end subroutine

end module fv3_sphere_mod
