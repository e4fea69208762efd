# the accumulations as load (top of the step) + add + store (where the value is ready) instead of L2 atomics
s = open('dsw_fused.h').read()
i = s.index("FV3_D void run_bf(int gid) const {")
j = s.index("FV3_D void run_general(int gid) const {")
b = s[i:j]
old = """      const In in = nxt;
      nxt = load_in(r < rlast ? r + 1 : rlast);
      const int j = r - 3, jf = r - 2;
      const int jc = j < jA ? jA : j, jfc = jf < jA ? jA : jf;   // rows of the (dropped) stores / zero additions of the warm-up steps
"""
new = """      const In in = nxt;
      const int j = r - 3, jf = r - 2;
      const int jc = j < jA ? jA : j, jfc = jf < jA ? jA : jf;   // rows of the (dropped) stores / zero additions of the warm-up steps
      const vd cx_o = vload(a.cx + oCX, (long)g.iCX(ilo, r), s.F), cy_o = vload(a.cy + oCY, (long)g.iCY(ilo, jfc), s.A);
      const vd mfx_o = vload(mfx, (long)g.iFX(ilo, jc), s.F), mfy_o = vload(mfy, (long)g.iFY(ilo, jc), s.C);
      nxt = load_in(r < rlast ? r + 1 : rlast);
"""
assert old in b
b = b.replace(old, new)
rep = {"vaccum_z(a.cx + oCX, iCX, sh.cx, s.F, mCX, on);": "vstore_b(a.cx + oCX, iCX, cx_o + sh.cx, mCX, on);",
       "vaccum_z(a.cy + oCY, iCY, sh.cy, s.A, mCY, on);": "vstore_b(a.cy + oCY, iCY, cy_o + sh.cy, mCY, on);",
       "vaccum_z(mfx, iFX, fxm, s.F, mOF, on);            // sw_core.F90:928-940": "vstore_b(mfx, iFX, mfx_o + fxm, mOF, on);",
       "vaccum_z(mfy, iFY0, fym0, s.C, mO, on);": "vstore_b(mfy, iFY0, mfy_o + fym0, mO, on);"}
for a, c in rep.items():
    assert a in b, a
    b = b.replace(a, c)
s = s[:i] + b + s[j:]
open('dsw_fused.h', 'w').write(s)
