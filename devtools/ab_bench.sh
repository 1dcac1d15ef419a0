#!/bin/bash
# same-box A/B of bench.py under (library, knobs, bench args): devtools/ab_bench.sh "lib|ENV=..,ENV=..|bench args" ...
for spec in "$@"; do
  IFS='|' read -r lib envs bargs <<< "$spec"
  e=${envs//,/ }
  L=$PWD/devtools/ab/lib$lib.so; [ "$lib" = "release" ] && L=$PWD/fedrann_amd/libfedrann_hip.so
  out=$(env FEDRANN_HIP_LIB=$L $e python bench.py --steps 4 --warmup 1 --no-compare --no-host-span --cpu-baseline-seconds 0 $bargs 2>/dev/null | python -c "import json,sys; r=json.load(sys.stdin); k=r['kernels_ms']; print('%.2f ms/step  prefilter %.2f rerank %.2f dedup %.2f frac %.3f uncert %s' % (r['ms_per_step'], k['knn_prefilter'], k['knn_rerank'], k['knn_dedup'], r['roofline']['frac'], r['uncertified_queries_last_step']))")
  echo "[$spec] $out"
done
