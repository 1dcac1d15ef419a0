#!/bin/bash
# development sweep: bench.py under debug knobs.  FDR_KNN_SHAPE picks a kernel shape (index into
# kShapes), FDR_KNN_NSEG the target split, FDR_KNN_DEBUG=1 skips the top-k slow path (WRONG results,
# timing only).  usage: knn_sweep.sh "shape nseg debug" ...
for cfg in "$@"; do
  set -- $cfg
  out=$(FDR_KNN_SHAPE=$1 FDR_KNN_NSEG=$2 FDR_KNN_DEBUG=${3:-0} python bench.py --steps 3 --warmup 1 --cpu-baseline-seconds 0 $BENCH_ARGS 2>/dev/null | python -c "import json,sys; r=json.load(sys.stdin); print('%.2f ms knn  %.1f TF  %.1f Mpairs/s ok=%s' % (r['kernels_ms']['knn_tile'], r['roofline']['achieved'], r['value']/1e6, r['config']['self_check']))")
  echo "shape=$1 nseg=$2 debug=${3:-0} : $out"
done
