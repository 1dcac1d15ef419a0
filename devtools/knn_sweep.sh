#!/bin/bash
# development sweep: bench.py under a few debug knobs (FDR_KNN_DEBUG=1 skips the top-k slow path,
# results are then WRONG -- timing only)
for cfg in "0 " "0 1" "0 5" "0 6" "0 9" "1 5"; do
  set -- $cfg
  out=$(FDR_KNN_DEBUG=$1 FDR_KNN_NSEG=${2:-} python bench.py --steps 3 --warmup 1 --cpu-baseline-seconds 0 "${@:3}" 2>/dev/null | python -c "import json,sys; r=json.load(sys.stdin); print('%.2f ms knn  %.1f TF  ok=%s' % (r['kernels_ms']['knn_tile'], r['roofline']['achieved'], r['config']['self_check']))")
  echo "debug=$1 nseg=${2:-auto} : $out"
done
