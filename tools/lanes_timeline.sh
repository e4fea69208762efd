#!/bin/bash
# usage (GPU box): bash tools/lanes_timeline.sh <tag> -- rocprofv3 --kernel-trace of a few C384 L127 pairs in two lanes: the start / end
# of every kernel of ONE pair by stream, and how much of the side stream's kernel time lies inside the main stream's
TAG=${1:-tl}
R=$PWD
mkdir -p gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
FV3_LANES_ONLY_TWO=1 NPX=385 NPZ=127 REPS=2 timeout 900 rocprofv3 --kernel-trace -d /tmp/ktl -- python $R/tools/lanes_check.py > /tmp/ktl.log 2>&1
cd $R
DB=$(find /tmp/ktl -name "*_results.db" | head -1)
python - $DB > gpurun_out/$TAG/lanes_timeline.txt <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if t.startswith("kernels") or "kernel_dispatch" in t]
# the documented view `kernels` has name, start, end, stream / queue ids
cols = [r[1] for r in con.execute("pragma table_info(kernels)")] if "kernels" in tabs else []
print("# tables:", [t for t in tabs if "kernel" in t][:8])
print("# columns of `kernels`:", cols)
sid = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
rows = list(con.execute(f"select name, start, end, {sid or 0} from kernels order by start"))
# the last pair: from the last kernel whose name has CswMarch back to ... simply take the last 60 kernels
rows = rows[-70:]
t0 = rows[0][1]
streams = sorted({r[3] for r in rows})
for n, s, e, q in rows:
    short = n.split("fv3::")[-1][:60] if "fv3::" in n else n[:60]
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} us  stream {streams.index(q)}  {short}")
# overlap of the streams
by = {}
for n, s, e, q in rows:
    by.setdefault(q, []).append((s, e))
if len(by) >= 2:
    qs = sorted(by, key=lambda q: -sum(e - s for s, e in by[q]))
    main, side = by[qs[0]], by[qs[1]]
    tot = sum(e - s for s, e in side)
    inside = 0
    for s, e in side:
        for a, b in main:
            inside += max(0, min(e, b) - max(s, a))
    print(f"# side stream: {tot / 1e6:.3f} ms of kernels, {inside / 1e6:.3f} ms of it while a kernel of the main stream runs ({100.0 * inside / max(tot, 1):.0f} %)")
PY
tail -3 gpurun_out/$TAG/lanes_timeline.txt
