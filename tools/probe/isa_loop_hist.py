"""Instruction histogram of a kernel's hottest loop from a hipcc -S listing.
usage: python tools/probe/isa_loop_hist.py fv3.s <mangled-kernel-substring> [--all]
Finds the kernel, takes its largest backward-branch loop body and prints the mnemonic histogram, grouped
(VALU f64 arithmetic, v_mov / v_accvgpr, DPP, v_cndmask, compares, SALU, VMEM, LDS, waitcnt)."""
import re, sys, collections


def kernel_body(path, key):
    out, on = [], False
    with open(path) as f:
        for line in f:
            if not on:
                if line.startswith("_Z") and key in line and line.rstrip().endswith(tuple(":")) or (line.startswith("_Z") and key in line and ": " in line):
                    on = True
                continue
            if line.startswith(".Lfunc_end"):
                break
            out.append(line.rstrip("\n"))
    return out


def loops(body):
    labels = {}
    for n, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = n
    res = []
    for n, l in enumerate(body):
        m = re.match(r"^\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"^\s+s_branch\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < n:
            res.append((labels[m.group(1)], n))
    return res


def classify(mn, line):
    if "dpp" in mn or " row_" in line or "wave_sh" in line or "quad_perm" in line:
        return "dpp"
    if mn.startswith("v_accvgpr"):
        return "accvgpr"
    if mn.startswith("v_mov") or mn.startswith("v_pk_mov"):
        return "v_mov"
    if mn.startswith("v_cndmask"):
        return "v_cndmask"
    if mn.startswith("v_cmp"):
        return "v_cmp"
    if mn.startswith("v_") and "f64" in mn:
        return "valu_f64"
    if mn.startswith("v_"):
        return "valu_other"
    if mn.startswith("s_waitcnt"):
        return "s_waitcnt"
    if mn.startswith("s_"):
        return "salu"
    if mn.startswith("global_") or mn.startswith("buffer_") or mn.startswith("flat_") or mn.startswith("scratch_"):
        return "vmem"
    if mn.startswith("ds_"):
        return "lds"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    body = kernel_body(path, key)
    if not body:
        sys.exit("kernel not found")
    ls = loops(body)
    ls.sort(key=lambda ab: ab[1] - ab[0], reverse=True)
    print("kernel lines", len(body), "loops (lines):", [(b - a) for a, b in ls[:8]])
    pick = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 0
    a, b = ls[pick]
    grp, mnem = collections.Counter(), collections.Counter()
    for l in body[a:b + 1]:
        m = re.match(r"^\s+([a-z_0-9]+)", l)
        if not m or l.strip().startswith((";", ".")):
            continue
        mn = m.group(1)
        g = classify(mn, l)
        grp[g] += 1
        mnem[(g, mn)] += 1
    tot = sum(grp.values())
    print("loop instructions:", tot)
    for g, c in grp.most_common():
        print(f"  {g:12s} {c:5d}")
        if "--all" in sys.argv:
            for (gg, mn), cc in sorted(mnem.items(), key=lambda kv: -kv[1]):
                if gg == g:
                    print(f"      {mn:28s} {cc}")
    for l in body:
        if "vgpr_count" in l or "NumVgprs" in l or "NumAgprs" in l or "ScratchSize" in l or "Occupancy" in l:
            print(l.strip())


main()
