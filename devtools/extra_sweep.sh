#!/bin/bash
# development sweep: prefilter candidate slack (FDR_KNN_EXTRA) -> kernel times and uncertified count
for ex in "$@"; do
  out=$(FDR_KNN_EXTRA=$ex python bench.py --steps 3 --warmup 1 --cpu-baseline-seconds 0 --no-compare --mode prefilter $BENCH_ARGS 2>/dev/null | python -c "import json,sys; r=json.load(sys.stdin); k=r['kernels_ms']; print('%.1f Mpairs/s  %.2f ms/step  prefilter %.2f  rerank %.2f  exact-fallback %.2f ms/launch  uncertified %s' % (r['value']/1e6, r['ms_per_step'], k['knn_prefilter'], k['knn_rerank'], k['knn_tile'], r['uncertified_queries_last_step']))")
  echo "extra=$ex : $out"
done
