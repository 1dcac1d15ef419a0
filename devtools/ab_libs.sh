#!/bin/bash
# A/B of whole libraries (release-mode builds that differ in compile-time switches): the same bench.py workload under each
# usage: bash devtools/ab_libs.sh "name name ..." "bench.py arguments"      (devtools/ab/libNAME.so)
for n in $1; do
  out=$(FEDRANN_HIP_LIB=$PWD/devtools/ab/lib$n.so python bench.py $2 --steps ${STEPS:-3} --warmup 1 --no-compare --no-host-span --cpu-baseline-seconds 0 2>gpurun_out/ab_$n.err | python -c "import json,sys; r=json.load(sys.stdin); k=r['kernels_ms']; print('%.2f ms/step  prefilter %.2f  rerank %.2f dedup %.2f frac %.3f  uncert %s' % (r['ms_per_step'], k['knn_prefilter'], k['knn_rerank'], k['knn_dedup'], r['roofline']['frac'], r['uncertified_queries_last_step']))" || tail -n 3 gpurun_out/ab_$n.err)
  echo "AB [$n | $2] $out"
  grep "fdr stamps" gpurun_out/ab_$n.err | tail -n 8
done
