s = open('csw_march.h').read()
s = s.replace("vstore_b(", "vstore_b_nt(")
open('csw_march.h', 'w').write(s)
