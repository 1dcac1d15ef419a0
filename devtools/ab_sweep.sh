#!/bin/bash
# development A/B: bench.py with alternative builds of the library (devtools/ab/lib*.so), both modes
for lib in "$@"; do
  for mode in prefilter exact; do
    out=$(FEDRANN_HIP_LIB=$PWD/devtools/ab/lib$lib.so python bench.py --steps 4 --warmup 1 --cpu-baseline-seconds 0 --no-compare --mode $mode $BENCH_ARGS 2>/dev/null | python -c "import json,sys; r=json.load(sys.stdin); k=r['kernels_ms']; print('%.1f Mpairs/s  %.2f ms/step  tile %.2f prefilter %.2f  ok=%s' % (r['value']/1e6, r['ms_per_step'], k['knn_tile'], k['knn_prefilter'], r['config']['self_check']))")
    echo "lib=$lib mode=$mode : $out"
  done
done
