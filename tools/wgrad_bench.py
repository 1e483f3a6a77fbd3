"""weight-gradient GEMM shapes: fp32 MFMA (k_gemm<*,false,false>) vs split-bf16 with transposing LDS reads (k_gemm_bf3_rc).
split-K slices are expressed as groups (same kernels, same block counts as the model's backward)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aread_amd import _lib as L
from tools.gemm_bench import timeit

SHAPES = [("expert L1  dW[1024,288]", 1024, 288, 9728, 1, 16), ("expert L2  dW[4][128,256]", 128, 256, 9728, 4, 32),
          ("expert L3  dW[4][64,128]", 64, 128, 9728, 4, 76), ("tower  l0  dW[3][64,64]", 64, 64, 9728, 3, 76)]
for name, M, N, K, G, ks in SHAPES:
    kc = (K // ks) // 64 * 64 or 64
    ks = K // kc
    A = torch.randn(K * G * M, device="cuda"); B = torch.randn(K * G * N, device="cuda")
    C = torch.empty(G * ks * M * N, device="cuda")
    # operands [K][G*M] / [K][G*N] (row-contiguous, groups side by side); slices along K as extra groups is not expressible
    # with one stride pair, so time G = 1 per slice-group: use G*ks groups over a [ks][kc][G*M] view
    lda, ldb = G * M, G * N
    def f32():
        for g in range(G):
            L.check(L.lib().aread_gemm(L.ptr(A) + 4 * g * M, lda, kc * lda, 0, L.ptr(B) + 4 * g * N, ldb, kc * ldb, 0, L.ptr(C) + 4 * g * ks * M * N,
                                       N, M * N, None, 0, M, N, kc, ks, 0, L.stream()))
    def bf3():
        for g in range(G):
            L.check(L.lib().aread_gemm_bf16x3_rc(L.ptr(A) + 4 * g * M, lda, kc * lda, L.ptr(B) + 4 * g * N, ldb, kc * ldb,
                                                 L.ptr(C) + 4 * g * ks * M * N, N, M * N, M, N, kc, ks, 0, L.stream()))
    t1, t3 = timeit(f32), timeit(bf3)
    fl = 2.0 * M * N * kc * ks * G
    print(f"{name:28s} k_split {ks:3d}: fp32 {t1:7.1f} us ({fl / t1 / 1e6:6.1f} TF)   split-bf16 rc {t3:7.1f} us ({fl / t3 / 1e6:6.1f} TF algorithmic)")
