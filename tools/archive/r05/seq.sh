#!/bin/bash
# kernel sequence of one steady forward (where do the small copy / fill launches come from?)
out=gpurun_out/r05i; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d $out/prof -o fw -- python tools/forward_loop.py --streams 1 --forwards 6 > $out/prof.log 2>&1
db=$(find $out/prof -name "*.db" | head -1)
python - "$db" <<'P'
import sqlite3, sys, re
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end from kernels order by start").fetchall()
# last forward: from the last-but-one 'enc_stem' pair to the end
idx = [i for i, r in enumerate(rows) if "enc_stem" in r[0]]
start = idx[-2]
seq = rows[start:]
def short(n):
    n = re.sub(r"\(.*", "", n); n = n.replace("void ", "")
    return n[:60]
out = []
prev = None; cnt = 0
for n, s, e in seq:
    sn = short(n)
    if sn == prev: cnt += 1
    else:
        if prev: out.append((prev, cnt))
        prev, cnt = sn, 1
out.append((prev, cnt))
# compress the GRU loop body
for n, c in out:
    print(f"{c:3d} x {n}")
P
find $out/prof -name "*.db" -delete
