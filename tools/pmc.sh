#!/bin/bash
# usage: tools/pmc.sh <outdir> <kernel-regex> <counter set> -- <command...>     (run on the GPU box via gpurun)
# One rocprofv3 --pmc pass (counters in their own run, no tracing flags), CSV output.
out=$1; re=$2; set=$3; shift 4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 rocprofv3 --pmc $set --kernel-include-regex "$re" -f csv -d "$out" -o x -- "$@" > /dev/null 2>&1
python - "$out" "$re" <<'PY'
import csv, sys, collections, glob, re
rows = []
for f in glob.glob(sys.argv[1] + "/**/x_counter_collection.csv", recursive=True) + glob.glob(sys.argv[1] + "/x_counter_collection.csv"):
    rows += list(csv.DictReader(open(f)))
agg, n = collections.defaultdict(float), collections.defaultdict(int)
for r in rows:
    if re.search(sys.argv[2], r["Kernel_Name"]):
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(agg): print(f"{k:40s} launches {n[k]:3d}  avg {agg[k] / n[k]:.6g}")
PY
