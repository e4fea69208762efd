#!/bin/bash
# usage (on the GPU box): bash tools/pmc_l1.sh <tag> -- how hard every kernel of the bench line leans on the L1 (texture path): cache-line
# accesses (TCP_TOTAL_CACHE_ACCESSES_sum, 64 B each), L1 -> L2 read requests, load / store instructions, against its duration
# -> gpurun_out/<tag>/pmc_l1.csv (kernel, avg_us, L1 accesses, TB/s through the L1, accesses per VMEM instruction)
TAG=${1:-vX}
R=$PWD
mkdir -p gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -d /tmp/pl1 -- python $R/bench.py --no-cpu --steps 2 --warmup 1 > /tmp/pl1.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/pl2 -- python $R/bench.py --no-cpu --steps 2 --warmup 1 > /tmp/pl2.log 2>&1
cd $R
python - > gpurun_out/$TAG/pmc_l1.csv <<'PY'
import glob, sqlite3, collections
c = collections.defaultdict(dict)
for db in glob.glob("/tmp/pl1/**/*_results.db", recursive=True):
    con = sqlite3.connect(db)
    for k, n, v, m in con.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"):
        c[k][n] = v
        c[k]["launches"] = m
for db in glob.glob("/tmp/pl2/**/*_results.db", recursive=True):
    con = sqlite3.connect(db)
    for n_, c_, t_, a_, p_ in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        c[n_]["avg_us"] = a_ / 1e3 if a_ > 1e5 else a_
        c[n_]["total_us"] = t_
print("kernel,launches,avg_us,l1_accesses,l1_TBps,l2_read_req,vmem_rd,vmem_wr,l1_accesses_per_vmem")
rows = []
for k, v in c.items():
    if "TCP_TOTAL_CACHE_ACCESSES_sum" not in v or "avg_us" not in v:
        continue
    us = v["avg_us"]
    acc = v["TCP_TOTAL_CACHE_ACCESSES_sum"]
    vm = v.get("SQ_INSTS_VMEM_RD", 0) + v.get("SQ_INSTS_VMEM_WR", 0)
    rows.append((v.get("total_us", 0), k, v["launches"], us, acc, acc * 64 / (us * 1e-6) / 1e12 if us else 0, v.get("TCP_TCC_READ_REQ_sum", 0), v.get("SQ_INSTS_VMEM_RD", 0), v.get("SQ_INSTS_VMEM_WR", 0), acc / vm if vm else 0))
for r in sorted(rows, reverse=True):
    print("\"%s\",%d,%.2f,%.0f,%.2f,%.0f,%.0f,%.0f,%.1f" % (r[1].replace("fv3::", "")[:110], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9]))
PY
