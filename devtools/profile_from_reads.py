"""cProfile of the from-reads pipeline (see bench_from_reads.py for the arguments): where stage 1's host time goes."""
import cProfile
import pstats
import runpy
import sys

sys.argv = ["bench_from_reads.py"] + sys.argv[1:]
cProfile.run("runpy.run_path('devtools/bench_from_reads.py', run_name='__main__')", "/tmp/from_reads.prof")
pstats.Stats("/tmp/from_reads.prof").sort_stats("cumtime").print_stats(45)
