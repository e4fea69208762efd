import json, sys
def load(p):
    l=[x for x in open(p) if x.strip().startswith('{')][-1]
    return json.loads(l)
a, b = load(sys.argv[1]), load(sys.argv[2])
for d, n in ((a, 'A'), (b, 'B')):
    cs = d['cubed_sphere']
    pf = cs['pair_one_face']
    print(n, 'pair_face ms', round(pf['ms'], 4), 'march', round(pf['marching_kernels_ms'], 4), 'pass', round(pf['pass_kernels_ms'], 4), 'sphere sypd', round(cs['sphere_one_gpu']['sypd'], 4), 'wall', round(cs['sphere_one_gpu']['wall_s_per_dt_atmos'], 5), 'config2', round(cs['config2_c96_l79_hydrostatic']['sypd'], 2))
ka, kb = a['cubed_sphere']['sphere_one_gpu_kernels_ms_per_dt_atmos'], b['cubed_sphere']['sphere_one_gpu_kernels_ms_per_dt_atmos']
print('sum', round(sum(ka.values()), 2), round(sum(kb.values()), 2))
for k in sorted(ka, key=lambda k: -ka[k]):
    if abs(ka[k] - kb.get(k, 0)) > 0.3: print('  ', k, ka[k], kb.get(k))
