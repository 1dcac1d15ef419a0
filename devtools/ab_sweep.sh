#!/bin/bash
# development A/B: bench.py with alternative builds of the library (devtools/ab/lib*.so)
for lib in "$@"; do
  for nq in 2 1; do
    out=$(FEDRANN_HIP_LIB=$PWD/devtools/ab/lib$lib.so FDR_KNN_NQ=$nq python bench.py --steps 3 --warmup 1 --cpu-baseline-seconds 0 2>/dev/null | python -c "import json,sys; r=json.load(sys.stdin); print('%.2f ms knn  %.1f TF  ok=%s' % (r['kernels_ms']['knn_tile'], r['roofline']['achieved'], r['config']['self_check']))")
    echo "lib=$lib NQ=$nq : $out"
  done
done
