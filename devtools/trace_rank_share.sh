#!/bin/bash
# kernel trace (per-kernel totals) of one rank's share of config 4 / 5 (devtools/rank_share_real.py)
# usage (on the GPU box): bash devtools/trace_rank_share.sh 4|5 [reads] [tag]
cfg=${1:-4}; reads=${2:-10000000}; tag=${3:-c$cfg}
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$tag -o trace -- python3 $GRAFT_REPO_ROOT/devtools/rank_share_real.py $cfg $reads 8 1 > $out/prof_$tag.json 2> $out/prof_$tag.err || { tail -5 $out/prof_$tag.err; exit 1; }
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$out/prof_$tag/trace_kernel_stats.csv")))
for r in rows[:22]:
    print("%-70s calls %5s total %10.2f ms avg %9.1f us  %5s%%" % (r["Name"][:70], r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"][:5]))
PY
