#!/bin/bash
# development: GPU parity tests (optional), then the 100 k and 1 M benches without the other-mode pass
# usage: bash devtools/quick.sh [notest] [bench args...]
if [ "$1" != "notest" ]; then
  python -m pytest tests -m gpu -x -q > gpurun_out/quick_tests.log 2>&1 || { tail -30 gpurun_out/quick_tests.log; exit 1; }
  tail -2 gpurun_out/quick_tests.log
else shift; fi
run() { name=$1; shift; python bench.py "$@" --cpu-baseline-seconds 0 --no-compare > gpurun_out/q_$name.json 2> gpurun_out/q_$name.err || { tail -5 gpurun_out/q_$name.err; return 1; }; python - <<PY
import json
j=json.load(open("gpurun_out/q_$name.json")); print("$name", round(j["value"]/1e6,1), "M pairs/s", round(j["ms_per_step"],2), "ms", "uncert", j["uncertified_queries_last_step"], "unique", j["unique_rows_searched"]["targets"], {k:round(v,2) for k,v in j["kernels_ms"].items() if v}, "frac", round(j["roofline"]["frac"],3))
PY
}
run 100k --reads 100000 --steps 10 --warmup 2 "$@" && run 1m --reads 1000000 --steps 4 --warmup 1 "$@"
