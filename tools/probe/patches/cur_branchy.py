s = open('dsw_fused.h').read()
i = s.index("      const vd cx_o = COURANT ?")
j = s.index("      nxt = load_in(r < rlast ? r + 1 : rlast);", i)
s = s[:i] + s[j:]
rep = {"vstore_b(a.cx + oCX, iCX, cx_o + sh.cx, mCX, on);": "if (on) vaccum(a.cx + oCX, iCX, sh.cx, s.lC0, lFx1);",
       "vstore_b(a.cy + oCY, iCY, cy_o + sh.cy, mCY, on);": "if (on) vaccum(a.cy + oCY, iCY, sh.cy, lY0, lY1);",
       "vstore_b(mfx, iFX, mfx_o + fxm, mOF, on);": "if (on) vaccum(mfx, iFX, fxm, oC0, oF1);",
       "vstore_b(mfy, iFY0, mfy_o + fym0, mO, on);": "if (on) vaccum(mfy, iFY0, fym0, oC0, oC1);"}
for a, b in rep.items():
    assert a in s, a
    s = s.replace(a, b)
open('dsw_fused.h', 'w').write(s)
