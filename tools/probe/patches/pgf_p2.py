# variant: NhPGradFused launched with three wavefronts per SIMD forced (tile_kernel_2w + tile_waves = 3)
s = open("fv3_api.hip").read()
a = '    return launch_p(c, "nh_p_grad", grid, NhPGradFused<TI, TJ>::lds_doubles, kf);'
assert a in s
open("fv3_api.hip", "w").write(s.replace(a, a.replace("launch_p(", "launch_p2(")))
