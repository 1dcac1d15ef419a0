#!/bin/bash
# development sweep: a rank's slice of 1 M rows (devtools/rank_slice_bench.py) and the 100 k-row set under planner knobs
# usage: bash devtools/slice_sweep.sh "ENV=..,ENV=.." ...   (dev build: devtools/ab/libdev.so)
export FEDRANN_HIP_LIB=$PWD/devtools/ab/libdev.so
for spec in "$@"; do
  envs=${spec//,/ }
  [ "$spec" = "base" ] && envs=""
  s=$(env $envs FDR_KNN_DEBUG=8 python devtools/rank_slice_bench.py ${READS:-1000000} ${RANKS:-8} 2>gpurun_out/slice_err.txt | tr '\n' ' ')
  plan=$(grep "fdr plan" gpurun_out/slice_err.txt | sort -u | cut -c1-150 | tr '\n' '|')
  echo "SLICE [$spec] $s"
  echo "      $plan"
  if [ -z "$NO100K" ]; then
  b=$(env $envs python bench.py --reads 100000 --steps 20 --warmup 3 --no-compare --no-host-span --cpu-baseline-seconds 0 2>/dev/null | python -c "import json,sys; r=json.load(sys.stdin); k=r['kernels_ms']; print('%.3f ms/step  prefilter %.3f  frac %.3f' % (r['ms_per_step'], k['knn_prefilter'], r['roofline']['frac']))")
  echo "100K  [$spec] $b"
  fi
done
