#!/bin/bash
# development sweep over the plan's per-workgroup fixed-cost estimate (FDR_KNN_OV, in tiles)
for ov in "$@"; do
  out=$(FDR_KNN_OV=$ov FDR_KNN_DEBUG=8 python bench.py --steps 3 --warmup 1 --cpu-baseline-seconds 0 $BENCH_ARGS 2>/tmp/ov_err.txt | python -c "import json,sys; r=json.load(sys.stdin); print('%.2f ms knn  %.1f TF  merge %.2f ms  %.1f Mpairs/s ok=%s' % (r['kernels_ms']['knn_tile'], r['roofline']['achieved'], r['kernels_ms']['knn_merge'], r['value']/1e6, r['config']['self_check']))")
  echo "ov=$ov : $out   $(grep 'fdr plan' /tmp/ov_err.txt | head -1 | cut -c1-150)"
done
