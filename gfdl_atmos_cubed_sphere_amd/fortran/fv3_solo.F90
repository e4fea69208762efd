!> Solo driver of the Fortran host (fv3_host_mod) on one doubly periodic tile: reads a state written with the
!> reference's array shapes, runs `nsteps` calls of the fv_dynamics k_split loop on the GPU through the C ABI, writes
!> the state back.  usage: fv3_solo <input file> <output file>   (both raw little-endian streams, see below)
!>
!> input : int32  nx, ny, npz, nq, n_split, k_split, nsteps, last_step, hydrostatic (bit 0; bit 1 inline_q, 2 remap_te, 3 use_cond,
!>                4 moist_kappa)
!>         real64 dx, dy, f0, bdt, ptop, d_con, d_ext, beta ; ak(npz+1), bk(npz+1)
!>         real64 u(isd:ied,jsd:jed+1,npz) v(isd:ied+1,jsd:jed,npz) w delp pt (isd:ied,jsd:jed,npz) delz(is:ie,js:je,npz)
!>                phis(isd:ied,jsd:jed) q(isd:ied,jsd:jed,npz,nq) [q_con, cappa (isd:ied,jsd:jed,npz) with use_cond / moist_kappa]
!> output: real64 u, v, w, delp, pt, delz, q in the same shapes
program fv3_solo
  use iso_c_binding
  use fv3_mi355x_mod
  use fv3_host_mod
  implicit none
  character(len=1024) :: fin, fout
  integer(c_int) :: nx, ny, npz, nq, n_split, k_split, nsteps, last_step, hydrostatic
  real(c_double) :: dx, dy, f0, bdt, ptop, d_con, d_ext, beta
  real(c_double), allocatable :: ak(:), bk(:), u(:,:,:), v(:,:,:), w(:,:,:), delp(:,:,:), pt(:,:,:), delz(:,:,:), phis(:,:)
  real(c_double), allocatable :: q(:,:,:,:)
  real(c_double), allocatable, target :: q_con(:,:,:), cappa(:,:,:)
  logical :: moist
  type(fv3_flags) :: fl
  type(fv3_atmos) :: at
  integer :: un, n, isd, ied, jsd, jed

  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  open(newunit=un, file=trim(fin), access='stream', form='unformatted', status='old')
  read(un) nx, ny, npz, nq, n_split, k_split, nsteps, last_step, hydrostatic
  read(un) dx, dy, f0, bdt, ptop, d_con, d_ext, beta
  allocate(ak(npz+1), bk(npz+1))
  read(un) ak, bk
  isd = 1 - 3; ied = nx + 3; jsd = 1 - 3; jed = ny + 3
  allocate(u(isd:ied, jsd:jed+1, npz), v(isd:ied+1, jsd:jed, npz), w(isd:ied, jsd:jed, npz), delp(isd:ied, jsd:jed, npz))
  allocate(pt(isd:ied, jsd:jed, npz), delz(nx, ny, npz), phis(isd:ied, jsd:jed), q(isd:ied, jsd:jed, npz, max(1, nq)))
  read(un) u, v, w, delp, pt, delz, phis
  if (nq > 0) read(un) q
  moist = iand(hydrostatic, 24_c_int) /= 0               ! bits 3, 4: use_cond, moist_kappa -- then q_con, cappa (A x npz) follow
  if (moist) then
    allocate(q_con(isd:ied, jsd:jed, npz), cappa(isd:ied, jsd:jed, npz))
    read(un) q_con, cappa
  end if
  close(un)

  fl%n_split = n_split; fl%k_split = k_split; fl%ptop = ptop
  fl%hydrostatic = iand(hydrostatic, 1_c_int) /= 0; fl%inline_q = iand(hydrostatic, 2_c_int) /= 0    ! bit 1: inline_q
  fl%remap_te = iand(hydrostatic, 4_c_int) /= 0                                                           ! bit 2: remap_te
  fl%d_con = d_con; fl%d_ext = d_ext; fl%beta = beta
  fl%use_cond = iand(hydrostatic, 8_c_int) /= 0; fl%moist_kappa = iand(hydrostatic, 16_c_int) /= 0
  if (moist) then    ! six water species in tracers 1 .. 6 (sphum, liq_wat, rainwat, ice_wat, snowwat, graupel); cv_vap, c_liq, c_ice of gfdl_mp
    fl%moist%nwat = 6; fl%moist%sphum = 1; fl%moist%liq_wat = 2; fl%moist%rainwat = 3; fl%moist%ice_wat = 4
    fl%moist%snowwat = 5; fl%moist%graupel = 6
    fl%moist%cv_vap = 3.d0 * 461.50d0; fl%moist%c_liq = 4.218d3; fl%moist%c_ice = 2.106d3
  end if
  call fv3_host_init(at, int(nx), int(ny), int(npz), int(nq), dx, dy, f0, fl, ak, bk)
  write(*,'(a,i0)') 'fv3_solo: gridstruct geometry mode ', fv3_grid_geom(at%ctx)
  if (nq > 0) then
    call fv3_host_upload(at, u, v, w, delp, pt, delz, phis, q)
  else
    call fv3_host_upload(at, u, v, w, delp, pt, delz, phis)
  end if
  if (moist) then
    call fv3_check(fv3_memcpy_h2d(at%ctx, at%q_con, c_loc(q_con), at%nA * npz * 8_c_size_t), 'q_con')
    call fv3_check(fv3_memcpy_h2d(at%ctx, at%cappa, c_loc(cappa), at%nA * npz * 8_c_size_t), 'cappa')
    call fv3_check(fv3_sync(at%ctx), 'fv3_sync')
  end if
  do n = 1, nsteps
    call fv3_fv_dynamics(at, bdt, last_step /= 0 .and. n == nsteps)
  end do
  if (nq > 0) then
    call fv3_host_download(at, u, v, w, delp, pt, delz, q)
  else
    call fv3_host_download(at, u, v, w, delp, pt, delz)
  end if
  call fv3_host_final(at)

  open(newunit=un, file=trim(fout), access='stream', form='unformatted', status='replace')
  write(un) u, v, w, delp, pt, delz
  if (nq > 0) write(un) q
  close(un)
  write(*,'(a,es24.16)') 'fv3_solo: done, sum(delp) = ', sum(delp(1:nx, 1:ny, :))
end program fv3_solo
