
"""Developer tool: the host <-> device transfers a Unity<T>-style caller pays around one image (pinned host memory):
u8 pixels in, Feature<SIFT_Descriptor>[F] out.  usage: python tools/bench_pcie.py [size] [features]"""
import sys
import time
import torch

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
F = int(sys.argv[2]) if len(sys.argv) > 2 else 1600000
pix_h = torch.empty(S * S, dtype=torch.uint8).pin_memory()
pix_d = torch.empty(S * S, dtype=torch.uint8, device="cuda")
feat_d = torch.empty(F * 152, dtype=torch.uint8, device="cuda")
feat_h = torch.empty(F * 152, dtype=torch.uint8).pin_memory()


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


h2d = timeit(lambda: pix_d.copy_(pix_h, non_blocking=True))
d2h = timeit(lambda: feat_h.copy_(feat_d, non_blocking=True))
print("H2D %dx%d u8 (%.1f MB): %.2f ms = %.1f GB/s" % (S, S, S * S / 1e6, h2d, S * S / h2d / 1e6))
print("D2H %d features (%.1f MB): %.2f ms = %.1f GB/s" % (F, F * 152 / 1e6, d2h, F * 152 / d2h / 1e6))
