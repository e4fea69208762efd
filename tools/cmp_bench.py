import json, sys
a = json.load(open(sys.argv[1])); b = json.load(open(sys.argv[2]))
print("pair ms", a["ms_per_step"], b["ms_per_step"])
print("model sypd", a["model_step"]["sypd"], b["model_step"]["sypd"], " sphere sypd", a["c384_sphere_one_gpu_sypd"], b["c384_sphere_one_gpu_sypd"], " cfg2", a["cubed_sphere"]["config2_c96_l79_hydrostatic"]["sypd"], b["cubed_sphere"]["config2_c96_l79_hydrostatic"]["sypd"])
for key in (("model_step", "kernels_ms_per_dt_atmos"), ("cubed_sphere", "sphere_one_gpu_kernels_ms_per_dt_atmos")):
    ka, kb = a[key[0]][key[1]], b[key[0]][key[1]]
    print(key[1], "sum", round(sum(ka.values()), 2), round(sum(kb.values()), 2))
    for n in ka:
        if abs(ka[n] - kb.get(n, 0)) > 0.03 * max(ka[n], 0.3):
            print("   %-22s %8.3f -> %8.3f" % (n, ka[n], kb.get(n, 0)))
