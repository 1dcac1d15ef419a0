#!/bin/bash
# round-4 experiment: the ping-pong candidate pass (FDR_KNN_PP=1) against round 3's shapes (FDR_KNN_PP=0), one development
# library, same box.  usage: bash devtools/exp_pp.sh LIB "workload;workload;..."   (a workload = bench.py arguments)
export FEDRANN_HIP_LIB=$PWD/devtools/ab/lib$1.so
IFS=';' read -ra WL <<< "$2"
for w in "${WL[@]}"; do
  for pp in 0 1; do
    for dbg in ${DBGS:-0}; do
      out=$(FDR_KNN_PP=$pp FDR_KNN_DEBUG=$dbg python bench.py $w --steps ${STEPS:-3} --warmup 1 --no-compare --no-host-span --cpu-baseline-seconds 0 2>gpurun_out/exp_err.log | python -c "import json,sys; r=json.load(sys.stdin); k=r['kernels_ms']; print('%.2f ms/step  prefilter %.2f  rerank %.2f dedup %.2f frac %.3f  unique %s launches %s q %s uncert %s' % (r['ms_per_step'], k['knn_prefilter'], k['knn_rerank'], k['knn_dedup'], r['roofline']['frac'], r['unique_rows_searched']['targets'], r['roofline']['launches_per_step'], r['roofline']['queues'], r['uncertified_queries_last_step']))" || tail -3 gpurun_out/exp_err.log)
      echo "PP [$w pp=$pp dbg=$dbg] $out"
    done
  done
done
