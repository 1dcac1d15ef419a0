#!/bin/bash
# one rocprofv3 --pmc pass over a short bench run; prints per-kernel sums of the counters
# usage: pmc_pass.sh NAME "COUNTER1 COUNTER2 ..." [bench args]
name=$1; ctrs=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$name -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-compare --cpu-baseline-seconds 0 "$@" > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/pmc_$name.err
cd $GRAFT_REPO_ROOT
python - <<PY
import sqlite3, collections
c = sqlite3.connect("gpurun_out/pmc_$name/p_results.db")
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
view = "counters_collection" if "counters_collection" in tabs else None
if view is None:
    print("tables:", tabs); raise SystemExit
cols = [r[1] for r in c.execute("pragma table_info(%s)" % view)]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
kn = "kernel_name" if "kernel_name" in cols else "name"
for row in c.execute("select %s, counter_name, value, dispatch_id from %s" % (kn, view)):
    k = row[0].split("(")[0][:50]
    acc[k][row[1]] += row[2]
seen = set()
for row in c.execute("select %s, dispatch_id from %s" % (kn, view)):
    if (row[0], row[1]) not in seen:
        seen.add((row[0], row[1])); cnt[row[0].split("(")[0][:50]] += 1
for k in acc:
    if any(t in k for t in ("prefilter", "merge_keys", "rerank_kernel", "embed_csr", "knn_tile")):
        print(k, "launches", cnt[k], {n: round(v / cnt[k]) for n, v in acc[k].items()})
PY
