#!/bin/bash
# development sweep for the d <= 256 prefilter shapes: bench.py on doubled rows, d = 256, k = 50 under knobs
# usage: bash devtools/d256_sweep.sh "ENV=..,ENV=.." ...   (dev build: devtools/ab/libdev.so; READS=500000)
export FEDRANN_HIP_LIB=$PWD/devtools/ab/libdev.so
for spec in "$@"; do
  envs=${spec//,/ }
  [ "$spec" = "base" ] && envs=""
  for dbg in ${DBGS:-0}; do
  b=$(env $envs FDR_KNN_DEBUG=$dbg python bench.py --reads ${READS:-500000} --doubling --dim ${DIM:-256} --knn ${KNN:-50} --steps 3 --warmup 1 --no-compare --no-host-span --cpu-baseline-seconds 0 2>/dev/null | python -c "import json,sys; r=json.load(sys.stdin); k=r['kernels_ms']; print('%.1f ms/step  prefilter %.1f  rerank %.1f dedup %.1f frac %.3f  unique %s launches %s q %s' % (r['ms_per_step'], k['knn_prefilter'], k['knn_rerank'], k['knn_dedup'], r['roofline']['frac'], r['unique_rows_searched']['targets'], r['roofline']['launches_per_step'], r['roofline']['queues']))")
  echo "D256 [$spec dbg=$dbg] $b"
  done
done
