#!/bin/bash
# A/B of prefilter workgroup shapes (FDR_KNN_PSHAPE: 0 = 4 waves, 8 = 8 waves, 16 = 8 waves x 2 query sets)
for cfg in "128 20" "256 50" "500 50"; do set -- $cfg; for r in 0 8 16; do for sl in 0 2; do
  if [ $sl = 0 ]; then unset FDR_KNN_SLOTS; else export FDR_KNN_SLOTS=$sl; fi
  FDR_KNN_PSHAPE=$r python bench.py --dim $1 --knn $2 --steps 3 --warmup 1 --cpu-baseline-seconds 0 --no-compare > gpurun_out/ring_$r.json 2> gpurun_out/ring_$r.err
  python - <<PY
import json
j=json.load(open("gpurun_out/ring_$r.json")); print("dim", $1, "pshape", $r, "slots", $sl, round(j["value"]/1e6,1), round(j["ms_per_step"],2), round(j["kernels_ms"]["knn_prefilter"],2), j["uncertified_queries_last_step"], j["config"]["self_check"])
PY
done; done; done
