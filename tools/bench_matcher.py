
"""Developer tool: stand-alone matcher timing (used under rocprofv3 for PMC passes)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import bench  # noqa: E402
from ssrlcv_amd import capi  # noqa: E402

torch.cuda.set_device(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
it = int(sys.argv[2]) if len(sys.argv) > 2 else 3
r = bench.bench_matcher(capi, torch, n, n, it)
print(n, "%.2f ms  %.0f TFLOP/s  frac %.3f" % (r["ms"], r["roofline"]["achieved"], r["roofline"]["frac"]))
if hasattr(capi.LIB, "ssrlcv_dbg_match_stats"):  # a -DSSRLCV_MATCH_STATS build: share of the chains that took the slow path
    import ctypes
    out = (ctypes.c_ulonglong * 12)()
    capi.LIB.ssrlcv_dbg_match_stats(out)
    print("chains %.1f M, through the slow path %.2f %%" % (out[3] / 1e6, 100.0 * out[4] / max(1, out[3])))
