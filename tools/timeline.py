
"""Kernel timeline of the last image of a `rocprofv3 --kernel-trace` run of bench.py (rocpd .db output): start, end,
duration (us), stream and kernel name, so that overlaps and waits between streams can be read off.
usage: timeline.py <results.db> [max_rows] [back]      back = which build_dog from the end (1 = the last one; bench.py ends with
14 stand-alone build_dog calls -- its stage_alone timing -- so the last fused image of its timed loop is back = 15)"""
import sqlite3
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:48]


def main():
    c = sqlite3.connect(sys.argv[1])
    lim = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    rows = c.execute("select name,start,end,stream_id from kernels order by start").fetchall()
    # every build_dog call starts with k_init_minmax
    ups = [i for i, r in enumerate(rows) if "k_init_minmax" in r[0]]
    back = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    i0 = ups[-back]
    t0 = rows[i0][1]
    for r in rows[i0:i0 + lim]:
        if "k_pack_i8" in r[0] or "k_match" in r[0]:
            break
        print("%9.1f %9.1f %8.1f s%-3d %s" % ((r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], short(r[0])))


if __name__ == "__main__":
    main()
