# timing reference only (wrong results): the branch-free transport kernel without its four accumulations
s = open('dsw_fused.h').read()
import re
n = 0
for a in ("vaccum_z(a.cx + oCX, iCX, sh.cx, s.F, mCX, on);", "vaccum_z(a.cy + oCY, iCY, sh.cy, s.A, mCY, on);",
          "vaccum_z(mfx, iFX, fxm, s.F, mOF, on);            // sw_core.F90:928-940", "vaccum_z(mfy, iFY0, fym0, s.C, mO, on);"):
    assert a in s, a
    s = s.replace(a, "")
open('dsw_fused.h', 'w').write(s)
