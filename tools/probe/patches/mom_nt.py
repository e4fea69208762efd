# the momentum kernel's and c_sw's stores with nt
s = open('dsw_fused.h').read()
i = s.index("struct DswMomentumFused {")
j = s.index("FV3_D void run_general(int gid) const {", i)
s = s[:i] + s[i:j].replace("vstore_b(", "vstore_b_nt(") + s[j:]
open('dsw_fused.h', 'w').write(s)
