#!/bin/bash
# usage (GPU box): bash tools/pmc_ab.sh <tag> <a.so:b.so> -- hardware counters of the pair's kernels for two builds in ONE process on ONE set of
# arrays (tools/pair_ab2.py): what a kernel waits for, what it issues, what it moves.  One --pmc group per pass (with --kernel-trace only).
TAG=${1:-pmc_ab}; SOS=$2
R=$PWD
mkdir -p gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
n=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum"; do
  n=$((n+1))
  FV3_AB_SO=$SOS timeout 600 rocprofv3 --pmc $C --kernel-trace -d /tmp/pab$n -- python $R/tools/pair_ab2.py 1 3 > /tmp/pab$n.log 2>&1
done
cd $R
python - > gpurun_out/$TAG/pmc_ab.csv <<'PY'
import glob, sqlite3, collections
# pair_ab2.py 1 3 launches every kernel 5 times per build (warm-up, build 0 then build 1), then 11 times per build: the dispatch order tells the build
print("kernel,build,counter,avg_per_launch,launches")
for db in sorted(glob.glob("/tmp/pab*/**/*_results.db", recursive=True)):
    con = sqlite3.connect(db)
    try:
        cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
        idc = "dispatch_id" if "dispatch_id" in cols else cols[0]
        rows = list(con.execute(f"select kernel_name, counter_name, value, {idc} from counters_collection order by {idc}"))
    except Exception as e:
        print("ERR", e); rows = []
    seq = collections.defaultdict(list)
    for k, c, v, i in rows:
        if "March" in k or "Fused" in k:
            seq[(k, c)].append(v)
    for (k, c), vs in seq.items():
        n = len(vs)
        if n != 32:
            print(f"\"{k[:60]}\",?,{c},{sum(vs)/n:.1f},{n}")
            continue
        b0 = vs[10:21]; b1 = vs[21:32]
        kk = k.split("fv3::")[2].split(">")[0] if k.count("fv3::") > 1 else k[:60]
        print(f"\"{kk}\",0,{c},{sum(b0)/len(b0):.1f},{len(b0)}")
        print(f"\"{kk}\",1,{c},{sum(b1)/len(b1):.1f},{len(b1)}")
PY
python - gpurun_out/$TAG/pmc_ab.csv <<'PY'
import sys, csv, collections
t = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    t[(r["kernel"], r["counter"])][r["build"]] = float(r["avg_per_launch"])
for (k, c), v in sorted(t.items()):
    a, b = v.get("0"), v.get("1")
    if a is not None and b is not None:
        print("%-36s %-24s %14.0f %14.0f  x%.3f" % (k[:36], c, a, b, b / a if a else 0))
PY
