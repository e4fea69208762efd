!> Solo driver of the Fortran host on the CUBED SPHERE (fv3_sphere_mod): the six faces in this one process (one library context per
!> face, every halo update through the cube-edge exchange behind the C ABI), `nsteps` calls of the fv_dynamics k_split loop.
!> usage: fv3_solo_sphere <input file> <output file>   (raw little-endian streams)
!>
!> input : int32  npx, npz, nq, n_split, k_split, nsteps, last_step, hydrostatic, nord
!>         real64 bdt, ptop, d_con, d_ext, da_min, da_min_c, d4_bg, beta ; ak(npz+1), bk(npz+1)
!>         per face (six times): the gridstruct members in the order of fv3_grid_host (A-, U-, V-, B-layout arrays with halo, sin_sg and
!>         cos_sg with 9 planes), then of fv3_grid_cubed: edge_w, edge_e, edge_s, edge_n (npx each), rsina (npx x npx), corner_f(12),
!>         a11, a12, a21, a22 (A), ec1, ec2 (A x 3), en1 ((npx-1) x npx x 3), en2 (npx x (npx-1) x 3);
!>         then the state: u, v, w, delp, pt (with halo, npz levels), delz ((npx-1)^2 x npz), phis (A), q (A x npz x nq)
!> output: per face u, v, w, delp, pt, delz, q in the same shapes
program fv3_solo_sphere
  use iso_c_binding
  use fv3_mi355x_mod
  use fv3_host_mod
  use fv3_sphere_mod
  implicit none
  character(len=1024) :: fin, fout
  integer(c_int) :: npx, npz, nq, n_split, k_split, nsteps, last_step, hydrostatic, nord
  real(c_double) :: bdt, ptop, d_con, d_ext, da_min, da_min_c, d4_bg, beta
  real(c_double), allocatable :: ak(:), bk(:)
  type gmet
    real(c_double), allocatable :: a(:,:,:), u(:,:,:), v(:,:,:), b(:,:,:), sg(:,:,:), cg(:,:,:)
    real(c_double), allocatable :: edge(:,:), rsina(:,:), a4(:,:,:), ec(:,:,:,:), en1(:,:,:), en2(:,:,:)
    real(c_double) :: corner_f(12)
  end type
  type fstate
    real(c_double), allocatable :: u(:,:,:), v(:,:,:), w(:,:,:), delp(:,:,:), pt(:,:,:), delz(:,:,:), phis(:,:), q(:,:,:,:)
  end type
  type(gmet), target :: gm(6)
  type(fstate), target :: st(6)
  type(fv3_flags) :: fl
  type(fv3_sphere) :: sp
  type(fv3_domain) :: dom
  type(fv3_grid_host) :: gh
  type(fv3_grid_cubed) :: gc
  integer :: un, n, t, nx, isd, ied

  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  open(newunit=un, file=trim(fin), access='stream', form='unformatted', status='old')
  read(un) npx, npz, nq, n_split, k_split, nsteps, last_step, hydrostatic, nord
  read(un) bdt, ptop, d_con, d_ext, da_min, da_min_c, d4_bg, beta
  allocate(ak(npz+1), bk(npz+1))
  read(un) ak, bk
  nx = npx - 1; isd = 1 - 3; ied = nx + 3
  do t = 1, 6
    allocate(gm(t)%a(isd:ied, isd:ied, 9), gm(t)%u(isd:ied, isd:ied+1, 9), gm(t)%v(isd:ied+1, isd:ied, 9), gm(t)%b(isd:ied+1, isd:ied+1, 4))
    allocate(gm(t)%sg(isd:ied, isd:ied, 9), gm(t)%cg(isd:ied, isd:ied, 9))
    allocate(gm(t)%edge(npx, 4), gm(t)%rsina(npx, npx), gm(t)%a4(isd:ied, isd:ied, 4), gm(t)%ec(isd:ied, isd:ied, 3, 2))
    allocate(gm(t)%en1(nx, npx, 3), gm(t)%en2(npx, nx, 3))
    read(un) gm(t)%a, gm(t)%u, gm(t)%v, gm(t)%b, gm(t)%sg, gm(t)%cg
    read(un) gm(t)%edge, gm(t)%rsina, gm(t)%corner_f, gm(t)%a4, gm(t)%ec, gm(t)%en1, gm(t)%en2
    allocate(st(t)%u(isd:ied, isd:ied+1, npz), st(t)%v(isd:ied+1, isd:ied, npz), st(t)%w(isd:ied, isd:ied, npz))
    allocate(st(t)%delp(isd:ied, isd:ied, npz), st(t)%pt(isd:ied, isd:ied, npz), st(t)%delz(nx, nx, npz), st(t)%phis(isd:ied, isd:ied))
    allocate(st(t)%q(isd:ied, isd:ied, npz, max(1, nq)))
    read(un) st(t)%u, st(t)%v, st(t)%w, st(t)%delp, st(t)%pt, st(t)%delz, st(t)%phis
    if (nq > 0) read(un) st(t)%q
  end do
  close(un)

  fl%n_split = n_split; fl%k_split = k_split; fl%ptop = ptop; fl%nord = nord; fl%d4_bg = d4_bg
  fl%hydrostatic = iand(hydrostatic, 1_c_int) /= 0; fl%inline_q = iand(hydrostatic, 2_c_int) /= 0    ! bit 1: inline_q
  fl%remap_te = iand(hydrostatic, 4_c_int) /= 0                                                           ! bit 2: remap_te
  fl%d_con = d_con; fl%d_ext = d_ext; fl%beta = beta
  dom%is = 1; dom%ie = nx; dom%js = 1; dom%je = nx; dom%ng = 3; dom%npx = npx; dom%npy = npx; dom%npz = npz; dom%grid_type = 0
  dom%do_diss_est = 0; dom%prevent_diss_cooling = 1; dom%stretched_grid = 0; dom%lim_fac = 1.d0
  do t = 1, 6
    gh%da_min = da_min; gh%da_min_c = da_min_c
    gh%area = c_loc(gm(t)%a(isd,isd,1)); gh%rarea = c_loc(gm(t)%a(isd,isd,2)); gh%dxa = c_loc(gm(t)%a(isd,isd,3))
    gh%dya = c_loc(gm(t)%a(isd,isd,4)); gh%rdxa = c_loc(gm(t)%a(isd,isd,5)); gh%rdya = c_loc(gm(t)%a(isd,isd,6))
    gh%cosa_s = c_loc(gm(t)%a(isd,isd,7)); gh%rsin2 = c_loc(gm(t)%a(isd,isd,8)); gh%f0 = c_loc(gm(t)%a(isd,isd,9))
    gh%dx = c_loc(gm(t)%u(isd,isd,1)); gh%rdx = c_loc(gm(t)%u(isd,isd,2)); gh%dyc = c_loc(gm(t)%u(isd,isd,3))
    gh%rdyc = c_loc(gm(t)%u(isd,isd,4)); gh%cosa_v = c_loc(gm(t)%u(isd,isd,5)); gh%sina_v = c_loc(gm(t)%u(isd,isd,6))
    gh%rsin_v = c_loc(gm(t)%u(isd,isd,7)); gh%divg_u = c_loc(gm(t)%u(isd,isd,8)); gh%del6_u = c_loc(gm(t)%u(isd,isd,9))
    gh%dy = c_loc(gm(t)%v(isd,isd,1)); gh%rdy = c_loc(gm(t)%v(isd,isd,2)); gh%dxc = c_loc(gm(t)%v(isd,isd,3))
    gh%rdxc = c_loc(gm(t)%v(isd,isd,4)); gh%cosa_u = c_loc(gm(t)%v(isd,isd,5)); gh%sina_u = c_loc(gm(t)%v(isd,isd,6))
    gh%rsin_u = c_loc(gm(t)%v(isd,isd,7)); gh%divg_v = c_loc(gm(t)%v(isd,isd,8)); gh%del6_v = c_loc(gm(t)%v(isd,isd,9))
    gh%rarea_c = c_loc(gm(t)%b(isd,isd,1)); gh%fC = c_loc(gm(t)%b(isd,isd,2)); gh%cosa = c_loc(gm(t)%b(isd,isd,3))
    gh%sina = c_loc(gm(t)%b(isd,isd,4))
    gh%sin_sg = c_loc(gm(t)%sg(isd,isd,1)); gh%cos_sg = c_loc(gm(t)%cg(isd,isd,1))
    gc%edge_w = c_loc(gm(t)%edge(1,1)); gc%edge_e = c_loc(gm(t)%edge(1,2)); gc%edge_s = c_loc(gm(t)%edge(1,3)); gc%edge_n = c_loc(gm(t)%edge(1,4))
    gc%rsina = c_loc(gm(t)%rsina(1,1)); gc%corner_f = gm(t)%corner_f
    gc%a11 = c_loc(gm(t)%a4(isd,isd,1)); gc%a12 = c_loc(gm(t)%a4(isd,isd,2)); gc%a21 = c_loc(gm(t)%a4(isd,isd,3)); gc%a22 = c_loc(gm(t)%a4(isd,isd,4))
    gc%ec1 = c_loc(gm(t)%ec(isd,isd,1,1)); gc%ec2 = c_loc(gm(t)%ec(isd,isd,1,2))
    gc%en1 = c_loc(gm(t)%en1(1,1,1)); gc%en2 = c_loc(gm(t)%en2(1,1,1))
    call fv3_sphere_init_face(sp, t, t - 1, dom, gh, gc, int(nq), fl, ak, bk)
    if (nq > 0) then
      call fv3_host_upload(sp%f(t), st(t)%u, st(t)%v, st(t)%w, st(t)%delp, st(t)%pt, st(t)%delz, st(t)%phis, st(t)%q)
    else
      call fv3_host_upload(sp%f(t), st(t)%u, st(t)%v, st(t)%w, st(t)%delp, st(t)%pt, st(t)%delz, st(t)%phis)
    end if
  end do
  call fv3_sphere_comm(sp, 0, 1, [0, 0, 0, 0, 0, 0])
  write(*,'(a,i0,a)') 'fv3_solo_sphere: six faces C', nx, ', halo updates through fv3_cube_halo_start / _complete'
  do n = 1, nsteps
    call fv3_sphere_fv_dynamics(sp, bdt, last_step /= 0 .and. n == nsteps, 1)
  end do
  open(newunit=un, file=trim(fout), access='stream', form='unformatted', status='replace')
  do t = 1, 6
    if (nq > 0) then
      call fv3_host_download(sp%f(t), st(t)%u, st(t)%v, st(t)%w, st(t)%delp, st(t)%pt, st(t)%delz, st(t)%q)
    else
      call fv3_host_download(sp%f(t), st(t)%u, st(t)%v, st(t)%w, st(t)%delp, st(t)%pt, st(t)%delz)
    end if
    write(un) st(t)%u, st(t)%v, st(t)%w, st(t)%delp, st(t)%pt, st(t)%delz
    if (nq > 0) write(un) st(t)%q
  end do
  close(un)
  call fv3_sphere_final(sp)
  write(*,'(a,es24.16)') 'fv3_solo_sphere: done, sum(delp face 1) = ', sum(st(1)%delp(1:nx, 1:nx, :))
end program fv3_solo_sphere
