"""Experiment: the two images of a bench step on two independent streams (plan k on stream k, no cross-stream events)
against the serial order of bench.py.  usage: python tools/bench_two_stream.py [steps]   (env SSRLCV_SIFT_SERIAL=1 etc. apply)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import helpers as _H  # noqa: E402
_H.limit_openmp()
import torch  # noqa: E402
from ssrlcv_amd import capi  # noqa: E402
import scene  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
W = 4096
dev = torch.device("cuda", 0)
imgs, _, _, _ = scene.pinhole_views(2, W, device=dev, seed=scene.SEED)
plans = [capi.SiftPlan(W, W) for _ in imgs]


def serial():
    for p, im in zip(plans, imgs):
        p.build_dog(im)
        p.describe()


streams = [torch.cuda.Stream() for _ in plans]


def two_streams():
    for p, im, s in zip(plans, imgs, streams):
        with torch.cuda.stream(s):
            p.build_dog(im)
            p.describe()


def staggered():
    # image 1's scale space beside image 0's key-point stage, nothing else overlapped inside a step
    with torch.cuda.stream(streams[0]):
        plans[0].build_dog(imgs[0])
    streams[1].wait_stream(streams[0])
    with torch.cuda.stream(streams[0]):
        plans[0].describe()
    with torch.cuda.stream(streams[1]):
        plans[1].build_dog(imgs[1])
    streams[0].wait_stream(streams[1])
    with torch.cuda.stream(streams[1]):
        streams[1].wait_stream(streams[0])
        plans[1].describe()


for name, fn in (("serial", serial), ("two_streams", two_streams), ("staggered", staggered), ("serial", serial)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("%-12s %.3f ms per step  %.0f Mpix/s   features %s" % (name, dt * 1e3, 2 * W * W / dt / 1e6, [p.count() for p in plans]), flush=True)
