"""Print the kernel timeline (start offset, duration, gap to previous) of the last bench step from a
rocprofv3 --kernel-trace results.db."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
rows = list(c.execute("select name, start, end from kernels order by start"))
# last step = after the last embed_csr kernel
last = max(i for i, r in enumerate(rows) if "embed_csr" in r[0])
t0 = rows[last][1]
prev_end = t0
for name, s, e in rows[last:]:
    short = name.split("(")[0][:60]
    print("%9.1f us  dur %8.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, short))
    prev_end = max(prev_end, e)
print("total %.1f us" % ((prev_end - t0) / 1e3))
