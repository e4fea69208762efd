#!/usr/bin/env python3
"""Turn rocprofv3 rocpd sqlite outputs (gpurun_out/prof_*/..._results.db) into the small text
summaries committed under profiles/.  usage: summarize_rocprof.py <tag> <stats.db> [<pmc.db> ...]"""
import json
import sqlite3
import sys


def stats(db):
    con = sqlite3.connect(db)
    rows = list(con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    out = ["name,calls,total_us,avg_us,percent"]
    for n, c, t, a, p in rows:
        out.append(f"\"{n}\",{c},{t:.3f},{a:.3f},{p:.2f}")
    return "\n".join(out)


def pmc(db):
    con = sqlite3.connect(db)
    rows = list(con.execute("select kernel_name,counter_name,avg(value),count(*) from counters_collection "
                            "group by kernel_name,counter_name"))
    return {(k, c): (v, n) for k, c, v, n in rows}


if __name__ == "__main__":
    tag = sys.argv[1]
    open(f"profiles/{tag}_kernel_stats.csv", "w").write(stats(sys.argv[2]) + "\n")
    allp = {}
    for db in sys.argv[3:]:
        allp.update(pmc(db))
    kernels = sorted({k for k, _ in allp})
    counters = sorted({c for _, c in allp})
    with open(f"profiles/{tag}_pmc.csv", "w") as f:
        f.write("kernel," + ",".join(counters) + "\n")
        for k in kernels:
            f.write("\"" + k + "\"," + ",".join(f"{allp[(k, c)][0]:.6g}" if (k, c) in allp else "" for c in counters) + "\n")
    print(open(f"profiles/{tag}_kernel_stats.csv").read())
    print(open(f"profiles/{tag}_pmc.csv").read())
