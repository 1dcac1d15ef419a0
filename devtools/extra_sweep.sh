#!/bin/bash
# candidates kept beyond k in the prefilter pass (FDR_KNN_EXTRA): step time and uncertified queries
for cfg in "100000 128 20" "1000000 128 20" "100000 128 50"; do set -- $cfg; for ex in 4 6 8 12; do
  FDR_KNN_EXTRA=$ex python bench.py --reads $1 --dim $2 --knn $3 --steps 3 --warmup 1 --cpu-baseline-seconds 0 --no-compare > gpurun_out/ex.json 2> gpurun_out/ex.err
  python - <<PY
import json
j=json.load(open("gpurun_out/ex.json")); print("reads", $1, "k", $3, "extra", $ex, round(j["value"]/1e6,1), "M/s", round(j["ms_per_step"],3), "ms  P1", round(j["kernels_ms"]["knn_prefilter"],3), "rerank", round(j["kernels_ms"]["knn_rerank"],3), "uncert", j["uncertified_queries_last_step"])
PY
done; done
