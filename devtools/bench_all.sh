#!/bin/bash
# the bench default + the other BASELINE / reference-default shapes, one JSON line each under gpurun_out/
run() { name=$1; shift; python bench.py "$@" --cpu-baseline-seconds 0 > gpurun_out/ba_$name.json 2> gpurun_out/ba_$name.err; python - <<PY
import json
j=json.load(open("gpurun_out/ba_$name.json")); print("$name", round(j["value"]/1e6,1), "M pairs/s", round(j["ms_per_step"],2), "ms", "uncert", j["uncertified_queries_last_step"], "unique", j["unique_rows_searched"]["targets"], {k:round(v,2) for k,v in j["kernels_ms"].items() if v}, "frac", round(j["roofline"]["frac"],3))
PY
}
run 100k --steps 5 --warmup 2 --no-compare
run 100k_k50 --knn 50 --steps 3 --warmup 1 --no-compare
run 100k_d256_k50 --dim 256 --knn 50 --steps 3 --warmup 1 --no-compare
run 100k_d500_k50 --dim 500 --knn 50 --doubling --steps 3 --warmup 1 --no-compare
run 1m --reads 1000000 --steps 2 --warmup 1 --no-compare
run 4m --reads 4000000 --steps 1 --warmup 1 --no-compare
