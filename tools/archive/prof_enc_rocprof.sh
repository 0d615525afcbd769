#!/bin/bash
# rocprofv3 kernel trace of the encoder engine alone (tools/archive/prof_enc.py fnet): true kernel durations without host pacing.
# usage (GPU box): tools/archive/prof_enc_rocprof.sh <outdir> [engine]
out=${1:-gpurun_out/enc_prof}; eng=${2:-pc}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d $out/prof -o enc -- python tools/archive/prof_enc.py $eng fnet > $out/prof.log 2>&1
db=$(find $out/prof -name "*.db" | head -1)
python tools/rocpd_summary.py "$db" $out/enc_kernel_stats.md > /dev/null 2>&1
python - "$db" > $out/enc_timeline.txt <<'P'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end from kernels order by start").fetchall()
# the last features() call: take the last 40 kernels
rows = rows[-40:]
prev = None
for n, s, e in rows:
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{(e - s) / 1e3:9.1f} us  gap {gap:7.1f} us  {n[:70]}")
    prev = e
P
find $out/prof -name "*.db" -delete; find $out/prof -name "*.csv" -size +2M -delete
cat $out/enc_timeline.txt
