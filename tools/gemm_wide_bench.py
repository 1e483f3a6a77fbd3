#!/usr/bin/env python
"""A/B of the expert-layer GEMM shapes: 64-row split-bf16 kernel (aread_gemm_bf16x3) vs the wide kernel
(aread_gemm_bf16x3_wide), interleaved rounds in one process on random data.  Usage (GPU box): python tools/gemm_wide_bench.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aread_amd  # noqa: E402
from aread_amd import _lib as L  # noqa: E402

SHAPES = [("expert L1 fwd", 9728, 1024, 288, 1), ("expert L2 fwd", 9728, 128, 256, 4), ("expert L3 fwd", 9728, 64, 128, 4),
          ("expert L1 dgrad", 9728, 288, 1024, 1), ("expert L2 dgrad", 9728, 256, 128, 4), ("expert L3 dgrad", 9728, 128, 64, 4),
          ("tower l0 L1", 9728, 64, 64, 3), ("tower l2 L2", 9728, 8, 16, 12)]


def timeit(fn, iters=20):
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        fn()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    lib = L.lib()
    only = os.environ.get("GEMM_ONLY")
    for name, M, N, K, G in SHAPES:
        M = int(os.environ.get("GEMM_M", M))
        if only and only not in name:
            continue
        A = torch.randn((M, G * K), device="cuda")
        W = torch.randn((G, N, K), device="cuda")
        bias = torch.randn((G, N), device="cuda")
        C = torch.empty((M, G * N), device="cuda")
        img = torch.empty(lib.aread_wimg_bytes(N, K, G), dtype=torch.uint8, device="cuda")
        L.check(lib.aread_wimg_prepare(L.ptr(W), N * K, K, 1, N, K, G, L.ptr(img), L.stream()))
        old = lambda: L.check(lib.aread_gemm_bf16x3(L.ptr(A), G * K, K, L.ptr(W), K, N * K, L.ptr(C), G * N, N, L.ptr(bias), N, M, N, K, G, 0, L.stream()))
        new = lambda: L.check(lib.aread_gemm_bf16x3_wide(L.ptr(A), G * K, K, L.ptr(img), L.ptr(C), G * N, N, L.ptr(bias), N, M, N, K, G, 0, L.stream()))
        prep = lambda: L.check(lib.aread_wimg_prepare(L.ptr(W), N * K, K, 1, N, K, G, L.ptr(img), L.stream()))
        for f in (old, new, prep):
            f()
        to, tn = [], []
        for _ in range(5):
            to.append(timeit(old)); tn.append(timeit(new))
        tp = timeit(prep)
        flops = 2.0 * M * N * K * G
        print(f"{name:18s} M={M} N={N} K={K} G={G}: old {np.median(to):7.2f} us  wide {np.median(tn):7.2f} us (min {min(tn):.2f})  "
              f"prep {tp:5.2f} us | wide: {flops / np.median(tn) / 1e6:7.1f} TFLOP/s algorithmic, issued frac {3 * flops / np.median(tn) / 1e6 / 2500:.3f}",
              flush=True)


if __name__ == "__main__":
    main()
