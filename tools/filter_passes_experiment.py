import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tools")
import numpy as np, torch
import helpers as H
H.limit_openmp()
from ssrlcv_amd import capi, pipeline
import scene
import test_gpu_configs as T
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
V = int(sys.argv[2]) if len(sys.argv) > 2 else 3
imgs, pbs, rig, sc = scene.pushbroom_views(V, S)
seed, _ = H.load_seed_features()
ws = pipeline.Workspace()
for passes in (0, 1, 2, 3, 4, 6):
    filt = [("statistical", 3.0, 0.1)] * passes
    res = pipeline.reconstruct(imgs, None, seed_features=seed, mode=0, pushbroom=pbs, ws=ws, filters=filt or None)
    mm, kp, pts = res["matches"], res["keypoints"], res["points"].cpu().numpy()
    err = T._ground_truth_error(rig, sc, mm, kp, pts)
    print("passes %d: %d of %d multi-matches kept, median %.4f km, within 0.2 km: %.1f %%  (%d bundles)  filter ms %.1f" % (
        passes, len(mm), res["matches_unfiltered"], np.median(err), 100 * (err < 0.2).mean(), int((err < 0.2).sum()),
        ws.times.get("filter", 0) * 1e3), flush=True)
    ws.times.clear()
