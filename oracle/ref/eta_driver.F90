!> Driver of the REFERENCE's own set_eta -- oracle/Makefile compiles the reference's stand-alone, FMS-free
!> docs/examples/FV3_level_transmogrifier/fv_eta.F90 (+ fv_eta.h) where it lies under /root/reference and links this
!> program against it (-> oracle/_ref/fv_eta_ref; test infrastructure, see oracle/fvo.h).  Prints ak(k), bk(k), k = 1..km+1.
!> usage: fv_eta_ref <km> [npz_type]
program eta_driver
  use fv_eta_mod
  implicit none
  integer :: km, k
  character(24) :: t
  character(64) :: arg
  real, allocatable :: ak(:), bk(:)
  call get_command_argument(1, arg)
  read(arg, *) km
  t = ' '
  if (command_argument_count() > 1) call get_command_argument(2, t)
  allocate(ak(km+1), bk(km+1))
  call set_eta(km, ak, bk, t)
  do k = 1, km + 1
    write(*, '(es26.18,1x,es26.18)') ak(k), bk(k)
  end do
end program eta_driver
