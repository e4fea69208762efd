# variant: NhPGradFused on 16 x 16 tiles
s = open("fv3_api.hip").read()
a = "constexpr int FI = 32, FJ = 8;"
assert a in s
open("fv3_api.hip", "w").write(s.replace(a, "constexpr int FI = 16, FJ = 16;"))
