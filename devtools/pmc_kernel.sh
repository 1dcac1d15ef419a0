#!/bin/bash
# per-launch average of counters for the kernels whose name contains SUBSTR, under a given library
# usage: FEDRANN_HIP_LIB=... bash devtools/pmc_kernel.sh NAME "COUNTERS" SUBSTR bench args...
name=$1; ctrs=$2; sub=$3; shift 3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmck_$name -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-compare --no-host-span --cpu-baseline-seconds 0 "$@" > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/pmck_$name.err
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
for f in glob.glob("gpurun_out/pmck_$name/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("<")[0][:40]
        if "$sub" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen: seen.add(key); n[k] += 1
for k in acc: print("PMC [$name]", k, n[k], {c: round(v / n[k]) for c, v in acc[k].items()})
PY
