s = open('dsw_fused.h').read()
i = s.index("FV3_D void run_bf(int gid) const {")
j = s.index("FV3_D void run_general(int gid) const {")
s = s[:i] + s[i:j].replace("vstore_b_nt(", "vstore_b(") + s[j:]
open('dsw_fused.h', 'w').write(s)
