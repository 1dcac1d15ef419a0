#!/bin/bash
# segments of the synchronised rounds (development knob FDR_KNN_COHORT_NS) against the cost model's choice (0)
# usage: bash devtools/ns_sweep.sh LIB "ns ns ..." "bench.py arguments"
export FEDRANN_HIP_LIB=$PWD/devtools/ab/lib$1.so
for ns in $2; do
  out=$(FDR_KNN_COHORT_NS=$ns python bench.py $3 --steps ${STEPS:-3} --warmup 1 --no-compare --no-host-span --cpu-baseline-seconds 0 2>gpurun_out/ns_err.log | python -c "import json,sys; r=json.load(sys.stdin); k=r['kernels_ms']; print('%.2f ms/step  prefilter %.2f  rerank %.2f frac %.3f launches %s uncert %s' % (r['ms_per_step'], k['knn_prefilter'], k['knn_rerank'], r['roofline']['frac'], r['roofline']['launches_per_step'], r['uncertified_queries_last_step']))" || tail -n 3 gpurun_out/ns_err.log)
  echo "NS [$3 ns=$ns] $out"
done
