#!/bin/bash
# SQ counters of the k-mer search kernels (one rocprofv3 --pmc pass over devtools/bench_kmer_search.py)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_kmer -o pmc -- python3 $GRAFT_REPO_ROOT/devtools/bench_kmer_search.py ${1:-20000} ${2:-10000} > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/pmc_kmer.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_kmer2 -o pmc -- python3 $GRAFT_REPO_ROOT/devtools/bench_kmer_search.py ${1:-20000} ${2:-10000} > /dev/null 2>> $GRAFT_REPO_ROOT/gpurun_out/pmc_kmer.err
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, collections
for d in ("pmc_kmer", "pmc_kmer2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open("gpurun_out/%s/pmc_counter_collection.csv" % d)):
        if "ks_search" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:20]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(k, {c: round(sum(v) / len(v)) for c, v in cs.items()})
PY
