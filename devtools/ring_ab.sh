#!/bin/bash
# A/B of prefilter stage depth (FDR_KNN_RING) x planner slots per CU at d=256 / d=500, k=50
for cfg in "256 50" "500 50"; do set -- $cfg; for r in 0 2; do for sl in 0 2 3; do
  if [ $sl = 0 ]; then unset FDR_KNN_SLOTS; else export FDR_KNN_SLOTS=$sl; fi
  FDR_KNN_DEBUG=8 FDR_KNN_RING=$r python bench.py --dim $1 --knn $2 --steps 3 --warmup 1 --cpu-baseline-seconds 0 --no-compare > gpurun_out/ring_$r.json 2> gpurun_out/ring_$r.err
  grep "fdr plan" gpurun_out/ring_$r.err | tail -1
  python - <<PY
import json
j=json.load(open("gpurun_out/ring_$r.json")); print("dim", $1, "ring", $r, "slots", $sl, j["value"], j["ms_per_step"], j["kernels_ms"]["knn_prefilter"], j["kernels_ms"]["knn_rerank"], j["uncertified_queries_last_step"])
PY
done; done; done
