# the branch-free transport kernel with the accumulations under lane / row conditions again (branches)
s = open('dsw_fused.h').read()
rep = {"vaccum_z(a.cx + oCX, iCX, sh.cx, s.F, mCX, on);": "if (on) vaccum(a.cx + oCX, iCX, sh.cx, s.lC0, lFx1);",
       "vaccum_z(a.cy + oCY, iCY, sh.cy, s.A, mCY, on);": "if (on) vaccum(a.cy + oCY, iCY, sh.cy, lY0, lY1);",
       "vaccum_z(mfx, iFX, fxm, s.F, mOF, on);            // sw_core.F90:928-940": "if (on) vaccum(mfx, iFX, fxm, oC0, oF1);",
       "vaccum_z(mfy, iFY0, fym0, s.C, mO, on);": "if (on) vaccum(mfy, iFY0, fym0, oC0, oC1);"}
for a, b in rep.items():
    assert a in s, a
    s = s.replace(a, b)
open('dsw_fused.h', 'w').write(s)
