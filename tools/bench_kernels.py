
"""Developer micro-benchmarks of individual C-ABI kernels (not part of the driver contract).
usage: python tools/bench_kernels.py [--size 4096]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ssrlcv_amd import capi  # noqa: E402
import ctypes  # noqa: E402


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=4096)
    args = ap.parse_args()
    W = H = args.size * 2  # octave-0 working size
    n = W * H
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    src = (torch.randn(n, device="cuda", generator=g) * 40 + 120).contiguous()
    print("octave-0 level %dx%d (%.1f Mpx, %.0f MB per level)" % (W, H, n / 1e6, n * 4 / 1e6))
    for sigma in (0.70710678, 1.0, 1.4142135, 2.0, 2.828427, 4.0):
        taps, wgt = capi.gauss_kernel(sigma, 0.5)
        ms = timeit(lambda: capi.gauss_sep_conv(src, W, H, wgt))
        print("gauss taps=%2d  %.3f ms  %.0f GB/s (8 B/px)  %.1f TFMA/s" % (taps, ms, n * 8 / ms / 1e6, n * 2 * taps / ms / 1e9))
    lv = [torch.empty(n, device="cuda") for _ in range(6)]
    for t in lv:
        t.copy_(src)
    dg = [torch.empty(n, device="cuda") for _ in range(5)]
    lmm = torch.tensor([0.0, 255.0] * 6, device="cuda")
    dmm = torch.tensor([3.4e38, -3.4e38] * 5, device="cuda")
    LV = (ctypes.c_void_p * 6)(*[t.data_ptr() for t in lv])
    DG = (ctypes.c_void_p * 5)(*[t.data_ptr() for t in dg])

    def dog():
        capi.check(capi.LIB.ssrlcv_hip_dog_normalised_sub(LV, capi.ptr(lmm), ctypes.c_uint32(W), ctypes.c_uint32(H), DG,
                                                          capi.ptr(dmm), capi.stream_ptr()))
    ms = timeit(dog)
    print("dog            %.3f ms  %.0f GB/s (44 B/px)" % (ms, n * 44 / ms / 1e6))

    def dog_nomm():
        capi.check(capi.LIB.ssrlcv_hip_dog_normalised_sub(LV, capi.ptr(lmm), ctypes.c_uint32(W), ctypes.c_uint32(H), DG,
                                                          ctypes.c_void_p(0), capi.stream_ptr()))
    ms = timeit(dog_nomm)
    print("dog (no minmax) %.3f ms  %.0f GB/s (44 B/px)" % (ms, n * 44 / ms / 1e6))
    taps, wgt = capi.gauss_kernel(4.0, 0.5)
    ms = timeit(lambda: capi.gauss_sep_conv(src, W, H, wgt, want_minmax=False))
    print("gauss taps=65 (no minmax) %.3f ms" % ms)
    taps, wgt = capi.gauss_kernel(0.70710678, 0.5)
    ms = timeit(lambda: capi.gauss_sep_conv(src, W, H, wgt, want_minmax=False))
    print("gauss taps=13 (no minmax) %.3f ms" % ms)
    ms = timeit(lambda: capi.bin2x(src, W, H))
    print("bin2x          %.3f ms  %.0f GB/s (5 B/px)" % (ms, n * 5 / ms / 1e6))
    img = torch.randint(0, 256, (args.size * args.size,), dtype=torch.uint8, device="cuda")
    ms = timeit(lambda: capi.upsample2x_u8(img, args.size, args.size))
    print("upsample_u8    %.3f ms  %.0f GB/s (4.25 B/px out)" % (ms, n * 4.25 / ms / 1e6))
    ms = timeit(lambda: lv[0].copy_(src))
    print("torch copy     %.3f ms  %.0f GB/s (8 B/px)" % (ms, n * 8 / ms / 1e6))


if __name__ == "__main__":
    main()
