#!/bin/bash
# one rocprofv3 --pmc pass over a short bench run of a development library; per-kernel averages incl. the clock the
# chip held (GRBM_GUI_ACTIVE / 8 / duration) and the matrix pipe's busy fraction
# usage: FEDRANN_HIP_LIB=... [FDR_KNN_*=...] bash devtools/pmc_shape.sh NAME "COUNTERS" bench args...
name=$1; ctrs=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$name -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-compare --no-host-span --cpu-baseline-seconds 0 "$@" > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/pmc_$name.err
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections
files = glob.glob("gpurun_out/pmc_$name/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.defaultdict(float)
seen = set()
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("<")[0][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key); n[k] += 1
            if "Start_Timestamp" in r: dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for k in acc:
    if any(t in k for t in ("prefilter", "range", "knn_tile")):
        a = {c: v / n[k] for c, v in acc[k].items()}
        d = dur[k] / n[k] if dur[k] else 0
        out = {c: round(v) for c, v in a.items()}
        if d and "GRBM_GUI_ACTIVE" in a:
            cyc = a["GRBM_GUI_ACTIVE"] / 8
            out["avg_us"] = round(d / 1e3, 1); out["clock_GHz"] = round(cyc / d, 3)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in a: out["mfma_busy"] = round(a["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), 3)
        if "SQ_WAVE_CYCLES" in a:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY"):
                if c in a: out[c + "_frac"] = round(a[c] / a["SQ_WAVE_CYCLES"], 3)
        print("PMC [$name]", k, "launches", n[k], out)
PY
