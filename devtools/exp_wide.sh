#!/bin/bash
# round-4 experiment: the wide stage body (knn_prefilter.inc: WIDE) against round 3's, same box
# usage: bash devtools/exp_wide.sh "lib lib ..."   (devtools/ab/lib<lib>.so, dev builds with FDR_SHAPE_MASK=0xC8 FDR_LH_MASK=32)
mkdir -p gpurun_out
bench() { # lib dbg args...
  lib=$1; dbg=$2; shift 2
  env FEDRANN_HIP_LIB=$PWD/devtools/ab/lib$lib.so FDR_KNN_DEBUG=$dbg python bench.py "$@" --steps 3 --warmup 1 --no-compare --no-host-span --cpu-baseline-seconds 0 2>gpurun_out/exp_err.log | python -c "import json,sys; r=json.load(sys.stdin); k=r['kernels_ms']; print('%.1f ms/step  prefilter %.1f  rerank %.1f dedup %.1f frac %.3f  unique %s launches %s q %s uncert %s' % (r['ms_per_step'], k['knn_prefilter'], k['knn_rerank'], k['knn_dedup'], r['roofline']['frac'], r['unique_rows_searched']['targets'], r['roofline']['launches_per_step'], r['roofline']['queues'], r['uncertified_queries_last_step']))" || tail -3 gpurun_out/exp_err.log
}
for lib in $1; do
  for dbg in ${DBGS:-0 1}; do
    echo "EXP [$lib dbg=$dbg d256k50 1M doubled] $(bench $lib $dbg --reads 500000 --doubling --dim 256 --knn 50)"
    echo "EXP [$lib dbg=$dbg d500k50 600k] $(bench $lib $dbg --reads 600000 --dim 500 --knn 50)"
  done
done
