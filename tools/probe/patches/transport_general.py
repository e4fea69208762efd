# the transport kernel in its general (round 4) form, everything else as it is
s = open('dsw_fused.h').read()
a = "    if constexpr (FLUXES || !FV3_BF)\n      run_general(gid);\n    else\n      run_bf(gid);"
assert a in s
s = s.replace(a, "    run_general(gid);")
open('dsw_fused.h', 'w').write(s)
s = open('fv3_api.hip').read()
a = "      if (FV3_BF && s0 == 0 && ns == NS && g0 == 0 && ng == NG && !c->tj_fixed && a.mask_w == 0)"
assert a in s
s = s.replace(a, "      if (false)")
a = "  lev_activate(c, (FV3_BF && c->sponge_march && g.geom == 2 && fused && march_m) ? 1 : 0);"
assert a in s
s = s.replace(a, "  lev_activate(c, 0);")
open('fv3_api.hip', 'w').write(s)
