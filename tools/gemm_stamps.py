#!/usr/bin/env python
"""Per-k-step time stamps of the wide split-bf16 GEMM (AREAD_GEMM_DBG=1): python tools/gemm_stamps.py M N K G"""
import os
import sys

os.environ["AREAD_GEMM_DBG"] = "1"
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aread_amd import _lib as L


def main():
    M, N, K, G = (int(a) for a in sys.argv[1:5])
    lib = L.lib()
    A = torch.randn((M, G * K), device="cuda")
    W = torch.randn((G, N, K), device="cuda")
    bias = torch.randn((G, N), device="cuda")
    C = torch.empty((M, G * N), device="cuda")
    img = torch.empty(lib.aread_wimg_bytes(N, K, G), dtype=torch.uint8, device="cuda")
    L.check(lib.aread_wimg_prepare(L.ptr(W), N * K, K, 1, N, K, G, L.ptr(img), L.stream()))
    for _ in range(5):
        L.check(lib.aread_gemm_bf16x3_wide(L.ptr(A), G * K, K, L.ptr(img), L.ptr(C), G * N, N, L.ptr(bias), N, M, N, K, G, 0, L.stream()))
    torch.cuda.synchronize()
    out = np.zeros((4, 256), np.uint64)
    L.check(lib.aread_debug_gemm_stamps(out.ctypes.data, out.size))
    ks = (K + 31) // 32
    for w in range(4):
        t = out[w].astype(np.int64)
        n = 1 + 3 * ks + 3
        if t[0] == 0:
            continue
        d = (t[:n] - t[0]) * 10      # ns at 100 MHz
        print(f"wg slot {w}: start {(t[0] - out[:, 0].astype(np.int64).min()) * 10} ns after the first; k-loop done {d[n-3]} ns, stores issued {d[n-2]}, stores acknowledged {d[n-1]}")
        for s in range(ks):
            a, b, c = d[1 + 3 * s: 4 + 3 * s]
            nxt = d[4 + 3 * s]
            print(f"   step {s:2d}: top {a:6d}  issue +{b - a:5d}  wait +{c - b:5d}  barrier +{nxt - c:5d}")


main()
