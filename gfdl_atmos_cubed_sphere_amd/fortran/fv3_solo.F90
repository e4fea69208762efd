!> Solo driver of the Fortran host (fv3_host_mod) on one doubly periodic tile: reads a state written with the
!> reference's array shapes, runs `nsteps` calls of the fv_dynamics k_split loop on the GPU through the C ABI, writes
!> the state back.  usage: fv3_solo <input file> <output file>   (both raw little-endian streams, see below)
!>
!> input : int32  nx, ny, npz, nq, n_split, k_split, nsteps, last_step, hydrostatic
!>         real64 dx, dy, f0, bdt, ptop, d_con, d_ext, beta ; ak(npz+1), bk(npz+1)
!>         real64 u(isd:ied,jsd:jed+1,npz) v(isd:ied+1,jsd:jed,npz) w delp pt (isd:ied,jsd:jed,npz) delz(is:ie,js:je,npz)
!>                phis(isd:ied,jsd:jed) q(isd:ied,jsd:jed,npz,nq)
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
  close(un)

  fl%n_split = n_split; fl%k_split = k_split; fl%ptop = ptop
  fl%hydrostatic = iand(hydrostatic, 1_c_int) /= 0; fl%inline_q = iand(hydrostatic, 2_c_int) /= 0    ! bit 1: inline_q
  fl%remap_te = iand(hydrostatic, 4_c_int) /= 0                                                           ! bit 2: remap_te
  fl%d_con = d_con; fl%d_ext = d_ext; fl%beta = beta
  call fv3_host_init(at, int(nx), int(ny), int(npz), int(nq), dx, dy, f0, fl, ak, bk)
  write(*,'(a,i0)') 'fv3_solo: gridstruct geometry mode ', fv3_grid_geom(at%ctx)
  if (nq > 0) then
    call fv3_host_upload(at, u, v, w, delp, pt, delz, phis, q)
  else
    call fv3_host_upload(at, u, v, w, delp, pt, delz, phis)
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
