#!/bin/bash
# Collect the per-round profile set on the GPU box (run through gpurun); then, here:
#   python devtools/summarize_prof.py r1      -> profiles/r1_kernel_stats.csv, profiles/r1_summary.json
# usage: bash devtools/profile_round.sh r1
tag=${1:-r1}
shift
extra="$@"   # extra bench.py arguments (e.g. --reads 1000000 --no-compare)
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --cpu-baseline-seconds 0 $extra"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$tag -o trace -- $B > $out/prof_$tag.bench.json 2> $out/prof_$tag.err || exit 1
# (the counter passes time nothing: no host-to-host passes, no other-mode pass -- every per-kernel figure of the summary
# then belongs to the timed workload's own launches)
P="python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-baseline-seconds 0 --no-host-span --no-compare $extra"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/prof_${tag}_fetch -o pmc -- $P > /dev/null 2> $out/prof_${tag}_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/prof_${tag}_write -o pmc -- $P > /dev/null 2> $out/prof_${tag}_write.err || exit 1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $out/prof_${tag}_sq -o pmc -- $P > /dev/null 2> $out/prof_${tag}_sq.err || exit 1
ls $out/prof_$tag $out/prof_${tag}_fetch
