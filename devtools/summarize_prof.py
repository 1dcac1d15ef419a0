#!/usr/bin/env python
"""Condense rocprofv3 output (gpurun_out/prof_<tag>*/) into profiles/<tag>_*.{csv,json}.

usage: python devtools/summarize_prof.py r1
  gpurun_out/prof_<tag>/trace_kernel_stats.csv            <- rocprofv3 --kernel-trace --stats
  gpurun_out/prof_<tag>_{fetch,write,sq}/pmc_counter_collection.csv  <- separate --pmc passes
HBM bytes follow MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in units of 1024 B; on gfx950
FETCH_SIZE counts wide (16 B/lane) coalesced reads at half their bytes, so it is doubled.
"""
import collections
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "prof_%s" % tag, "trace_kernel_stats.csv"),
            os.path.join(dst, "%s_kernel_stats.csv" % tag))


def short(name):
    for k in ("knn_prefilter_pp_kernel", "knn_range_pp_kernel", "knn_tile_kernel", "knn_merge_keys_kernel", "knn_merge_kernel", "embed_csr_kernel",
              "normalize_rows_kernel", "pack_zero_bits_kernel", "knn_prefilter_kernel", "knn_rerank_kernel",
              "to_half_kernel", "gather_queries_kernel", "scatter_results_kernel", "knn_range_kernel",
              "hash_rows_kernel", "dedup_probe_kernel", "expand_classes_kernel", "zero_answer_kernel"):
        if k in name:
            return k
    return None


out = collections.defaultdict(dict)
for sub in ("fetch", "write", "sq"):
    path = os.path.join(src, "prof_%s_%s" % (tag, sub), "pmc_counter_collection.csv")
    if not os.path.exists(path):
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        for c, vals in cs.items():
            out[k][c] = sum(vals) / len(vals)
for k, c in out.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        c["hbm_read_bytes_corrected"] = 2.0 * c["FETCH_SIZE"] * 1024.0
        c["hbm_write_bytes"] = c["WRITE_SIZE"] * 1024.0
        c["hbm_bytes_per_launch"] = c["hbm_read_bytes_corrected"] + c["hbm_write_bytes"]
    if "GRBM_GUI_ACTIVE" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c and c["GRBM_GUI_ACTIVE"] > 0:
        cycles = c["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
        c["mfma_pipe_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * 1024.0)  # 1024 SIMDs
stats = {}
for r in csv.DictReader(open(os.path.join(dst, "%s_kernel_stats.csv" % tag))):
    k = short(r["Name"])
    if k:
        stats[k] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "pct": float(r["Percentage"])}
bench_args, workload = None, None
try:  # the bench line of the kernel-trace run names the workload the counters belong to
    line = [l for l in open(os.path.join(src, "prof_%s.bench.json" % tag)).read().splitlines() if l.startswith("{")][-1]
    cfg = json.loads(line)
    c = cfg["config"]
    bench_args = {"reads": c["reads"], "dim": c["dim"], "knn": c["knn"], "gpus": cfg["n_gpus"],
                  "doubling": bool(c.get("doubling", c["rows"] != c["reads"]))}
    workload = c["workload"]
    shutil.copy(os.path.join(src, "prof_%s.bench.json" % tag), os.path.join(dst, "%s_bench.json" % tag))
except Exception as e:
    print("no bench line:", e)
json.dump({"tag": tag, "workload": workload, "bench_args": bench_args,
           "kernel_stats": stats, "pmc_per_launch_avg": out}, open(os.path.join(dst, "%s_summary.json" % tag), "w"),
          indent=1, sort_keys=True)
print(json.dumps({"kernel_stats": stats, "pmc": {k: {c: v for c, v in cs.items() if "bytes" in c or "frac" in c}
                                                  for k, cs in out.items()}}, indent=1))
