#!/bin/bash
# kernel trace of a rank's slice (devtools/rank_slice_bench.py): per-kernel totals of the run
# usage (on the GPU box): bash devtools/trace_slice.sh [reads] [ranks]
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_slice -o trace -- python3 $GRAFT_REPO_ROOT/devtools/rank_slice_bench.py ${1:-1000000} ${2:-8} > $out/prof_slice.txt 2> $out/prof_slice.err || { tail -5 $out/prof_slice.err; exit 1; }
cat $out/prof_slice.txt
