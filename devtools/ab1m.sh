#!/bin/bash
# same-box A/B at 1 M rows: devtools/ab1m.sh "lib:ENV=..,ENV=.." ...   (lib = devtools/ab/lib<lib>.so); DBGS="0 1" debug values
for spec in "$@"; do
  lib=${spec%%:*}; envs=${spec#*:}; [ "$envs" = "$spec" ] && envs=""
  for dbg in ${DBGS:-0 1}; do
    out=$(env FEDRANN_HIP_LIB=$PWD/devtools/ab/lib$lib.so FDR_KNN_DEBUG=$dbg ${envs//,/ } python bench.py --reads ${READS:-1000000} --steps 3 --warmup 1 --no-compare --no-host-span --cpu-baseline-seconds 0 2>/dev/null | python -c "import json,sys; r=json.load(sys.stdin); k=r['kernels_ms']; print('%.2f ms/step  prefilter %.2f  frac %.3f' % (r['ms_per_step'], k['knn_prefilter'], r['roofline']['frac']))")
    echo "lib=$lib $envs debug=$dbg : $out"
  done
done
