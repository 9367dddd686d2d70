
"""Developer tool: double-constrained (epipolar band) matching vs brute force on the same synthetic sets.
usage: python tools/bench_matcher_band.py [N] [size]  -- N features per image, image edge `size` (fixture cameras are
for 1024 px; locations are drawn uniformly in the image)"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
import helpers as H  # noqa: E402
from ssrlcv_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
size = float(sys.argv[2]) if len(sys.argv) > 2 else 1024.0
q, t = bench.synth_descriptors(n, 1), bench.synth_descriptors(n, 2)
rng = np.random.default_rng(5)
q["loc"] = rng.uniform(0, size, (n, 2)).astype(np.float32)
t["loc"] = rng.uniform(0, size, (n, 2)).astype(np.float32)
cams = H.load_view("Pipeline2View")["cameras"]
proj = capi.projection_matrix(cams[1:2])
q_d, t_d = capi.to_dev(q), capi.to_dev(t)
ws = capi.match_workspace(n, n)
out = capi.dev_bytes(n * 48)
for mode, label in ((0, "brute force"), (1, "double constrained eps=25 delta=5")):
    params = capi.make_match_params(mode, 0, 1, 25.0, 5.0, 0.6, 3e7, cams[0:1], proj)
    capi.match(q_d, n, t_d, n, params, capi.OUT_DMATCH, workspace=ws, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        capi.match(q_d, n, t_d, n, params, capi.OUT_DMATCH, workspace=ws, out=out)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 2 * 1e3
    res = capi.to_host(out, H.DMATCH, n)
    print("%-36s N=%d  %.2f ms  valid %d" % (label, n, ms, int((res["invalid"] == 0).sum())))
