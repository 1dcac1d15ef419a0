#!/bin/bash
# development sweep: prefilter LDS ring geometry (FDR_KNN_RING = stages*10 + tiles per stage)
for ring in "$@"; do
  out=$(FDR_KNN_RING=$ring python bench.py --steps 3 --warmup 1 --cpu-baseline-seconds 0 --no-compare --mode prefilter $BENCH_ARGS 2>/dev/null | python -c "import json,sys; r=json.load(sys.stdin); k=r['kernels_ms']; print('%.1f Mpairs/s  %.2f ms/step  prefilter %.2f  ok=%s' % (r['value']/1e6, r['ms_per_step'], k['knn_prefilter'], r['config']['self_check']))")
  echo "ring=$ring : $out"
done
