# current branch-free transport kernel with the flux capacitors as L2 atomics (+0.0 on the masked lanes) instead of load-add-store
s = open('dsw_fused.h').read()
i = s.index("      const vd cx_o = COURANT ?")
j = s.index("      nxt = load_in(r < rlast ? r + 1 : rlast);", i)
s = s[:i] + s[j:]
rep = {"vstore_b(a.cx + oCX, iCX, cx_o + sh.cx, mCX, on);": "vaccum_z(a.cx + oCX, iCX, sh.cx, s.F, mCX, on);",
       "vstore_b(a.cy + oCY, iCY, cy_o + sh.cy, mCY, on);": "vaccum_z(a.cy + oCY, iCY, sh.cy, s.A, mCY, on);",
       "vstore_b(mfx, iFX, mfx_o + fxm, mOF, on);": "vaccum_z(mfx, iFX, fxm, s.F, mOF, on);",
       "vstore_b(mfy, iFY0, mfy_o + fym0, mO, on);": "vaccum_z(mfy, iFY0, fym0, s.C, mO, on);"}
for a, b in rep.items():
    assert a in s, a
    s = s.replace(a, b)
open('dsw_fused.h', 'w').write(s)
