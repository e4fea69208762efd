s = open('spmd.h').read()
for a in ("(p + off)[l] = x;", "unsafeAtomicAdd(p + off + l, x);", "__builtin_nontemporal_store(x, p + off + l);"):
    assert ("if (l >= lmin && l <= lmax) " + a) in s
    s = s.replace("if (l >= lmin && l <= lmax) " + a, "if (l >= lmin && l <= lmax && x == 1.2345e300) " + a)
open('spmd.h', 'w').write(s)
