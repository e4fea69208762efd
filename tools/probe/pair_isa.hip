// ISA probe: only the three marching kernels of the headline pair (uniform metrics, hord 10, NH), device-only -S in ~30 s.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-honor-nans --cuda-device-only -S tools/probe/pair_isa.hip -o /tmp/asm/pair.s
#include "../../gfdl_atmos_cubed_sphere_amd/csrc/csw_march.h"
#include "../../gfdl_atmos_cubed_sphere_amd/csrc/dsw_fused.h"
#include "../../gfdl_atmos_cubed_sphere_amd/csrc/fv3_launch.h"
using namespace fv3;
template <class F>
void inst(const F &f) { hipLaunchKernelGGL(wave_kernel<F>, dim3(1), dim3(kNT), 0, 0, f, 1, 0); }
template <class F>
void inst2(const F &f) { hipLaunchKernelGGL(wave_kernel_2w<F>, dim3(1), dim3(kNT), 0, 0, f, 1, 0); }
template <class F>
void inst3(const F &f) { hipLaunchKernelGGL(wave_kernel_3w<F>, dim3(1), dim3(kNT), 0, 0, f, 1, 0); }
void probe_all() {
#ifndef PROBE_NO_T
  inst2(DswTransportFused<10, true, true, 2>{});
#endif
#ifndef PROBE_NO_M
#ifdef PROBE_M3
  inst3(DswMomentumFused<8, 10, 2>{});
#else
  inst2(DswMomentumFused<8, 10, 2>{});
#endif
#endif
#ifndef PROBE_NO_C
  inst(CswMarch<1, 2>{});
#endif
}
