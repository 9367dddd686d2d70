"""Dump every kernel of a rocprofv3 --kernel-trace rocpd database as TSV: start_us end_us dur_us stream name
usage: trace_dump.py <results.db> > out.tsv"""
import sqlite3
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:60]


c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name,start,end,stream_id from kernels order by start").fetchall()
t0 = rows[0][1]
for r in rows:
    print("%.1f\t%.1f\t%.1f\ts%d\t%s" % ((r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], short(r[0])))
