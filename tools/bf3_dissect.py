"""split-bf16 GEMM: time vs K at the expert-L1 shape (fixed cost = launch + prologue + epilogue; slope = k-loop)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aread_amd import _lib as L
from tools.gemm_bench import timeit

M, N = 9728, 1024
for K in (32, 64, 128, 288, 576, 1024, 2048):
    A = torch.randn(M * K, device="cuda"); B = torch.randn(N * K, device="cuda"); C = torch.empty(M * N, device="cuda")
    f3 = lambda: L.check(L.lib().aread_gemm_bf16x3(L.ptr(A), K, M * K, L.ptr(B), K, N * K, L.ptr(C), N, M * N, None, 0, M, N, K, 1, 0,
                                                   L.stream()))
    f1 = lambda: L.check(L.lib().aread_gemm(L.ptr(A), K, M * K, 1, L.ptr(B), K, N * K, 1, L.ptr(C), N, M * N, None, 0, M, N, K, 1, 0,
                                            L.stream()))
    t3, t1 = timeit(f3), timeit(f1)
    print(f"K={K:5d}  bf3 {t3:7.1f} us ({6.0 * M * N * K / t3 / 1e6:7.1f} TF issued)   f32 {t1:7.1f} us ({2.0 * M * N * K / t1 / 1e6:6.1f} TF)")
C2 = torch.empty(M * N, device="cuda")
print(f"copy of C-sized buffer (39.8 MB): {timeit(lambda: C2.copy_(C)):.1f} us")
