#!/bin/bash
# embed / normalise kernel times under each of several whole libraries: bash devtools/ab_embed.sh "name name" "bench args"
for n in $1; do
  FEDRANN_HIP_LIB=$PWD/devtools/ab/lib$n.so python bench.py $2 --steps ${STEPS:-10} --warmup 2 --no-compare --no-host-span --cpu-baseline-seconds 0 2>gpurun_out/abe_$n.err | python -c "import json,sys; r=json.load(sys.stdin); k=r['kernels_ms']; print('AB [$n | $2] %.2f ms/step embed %.4f normalize %.4f prefilter %.2f' % (r['ms_per_step'], k['embed_csr'], k['normalize_rows'], k['knn_prefilter']))" || tail -n 3 gpurun_out/abe_$n.err
done
