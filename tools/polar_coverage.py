"""Developer experiment: what fraction of the gradient tables' 256 x 16 tiles does a key point's sampling window touch?
(k_polar builds every tile of the three DoG levels of every octave; a tile mask would skip the untouched ones.)
usage: python tools/polar_coverage.py [--size 4096] [--tile 256x16]"""
import argparse
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import helpers as H  # noqa: E402
from ssrlcv_amd import capi  # noqa: E402
import scene  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--tile", default="256x16", help="tile width x height in table entries")
    args = ap.parse_args()
    S = args.size
    TWd, THt = [int(v) for v in args.tile.split('x')]
    img = scene.pinhole_views(1, S)[0][0]
    plan = capi.SiftPlan(S, S)
    plan.build_dog(img)
    plan.set_stop_stage(5)  # the lists the sampling kernels see (before the orientation copies)
    plan.describe()
    tot_tiles = tot_hit = 0
    for o in range(4):
        kps, idx, _ = plan.keypoints(o, H.SSKEYPOINT)
        w = (2 * S) >> o
        pw = 0.5 * (1 << o)
        tx, ty = (w + TWd - 1) // TWd, (w + THt - 1) // THt
        for b in (1, 2, 3):
            k = kps[kps["blur"] == b]
            r = np.ceil(np.ceil(k["sigma"] * 6.0 / pw) * 1.4143) + 2.0
            x0 = np.clip(np.floor((k["loc"][:, 0] - r) / TWd).astype(int), 0, tx - 1)
            x1 = np.clip(np.floor((k["loc"][:, 0] + r) / TWd).astype(int), 0, tx - 1)
            y0 = np.clip(np.floor((k["loc"][:, 1] - r) / THt).astype(int), 0, ty - 1)
            y1 = np.clip(np.floor((k["loc"][:, 1] + r) / THt).astype(int), 0, ty - 1)
            m = np.zeros((ty + 1, tx + 1), np.int32)
            np.add.at(m, (y0, x0), 1)
            np.add.at(m, (y1 + 1, x0), -1)
            np.add.at(m, (y0, x1 + 1), -1)
            np.add.at(m, (y1 + 1, x1 + 1), 1)
            cov = (np.cumsum(np.cumsum(m, 0), 1)[:ty, :tx] > 0)
            print("octave %d level %d: %7d key points, mean radius %5.1f px, tiles touched %6d of %6d (%.1f %%)" %
                  (o, b, len(k), float(r.mean()) if len(k) else 0.0, int(cov.sum()), tx * ty, 100.0 * cov.mean()))
            tot_tiles += tx * ty   # (every tile is the same number of table entries of its level)
            tot_hit += int(cov.sum())
    print("all levels: %.1f %% of the tiles (= of the table entries) are touched" % (100.0 * tot_hit / tot_tiles))


if __name__ == "__main__":
    main()
