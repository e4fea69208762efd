# the flux-capacitor sums stored with nt as well
s = open('dsw_fused.h').read()
for a in ("vstore_b(a.cx + oCX, iCX, cx_o + sh.cx, mCX, on);", "vstore_b(a.cy + oCY, iCY, cy_o + sh.cy, mCY, on);", "vstore_b(mfx, iFX, mfx_o + fxm, mOF, on);",
          "vstore_b(mfy, iFY0, mfy_o + fym0, mO, on);"):
    assert a in s, a
    s = s.replace(a, a.replace("vstore_b(", "vstore_b_nt("))
open('dsw_fused.h', 'w').write(s)
