"""split-bf16 GEMM (N=1024, K=288): time vs number of workgroups -- latency-bound or throughput-bound?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aread_amd import _lib as L
from tools.gemm_bench import timeit
N, K = 1024, 288
for M in (64, 640, 1216, 2432, 4864, 9728, 19456, 38912):
    A = torch.randn(M * K, device="cuda"); B = torch.randn(N * K, device="cuda"); C = torch.empty(M * N, device="cuda")
    f3 = lambda: L.check(L.lib().aread_gemm_bf16x3(L.ptr(A), K, M * K, L.ptr(B), K, N * K, L.ptr(C), N, M * N, None, 0, M, N, K, 1, 0, L.stream()))
    t = timeit(f3)
    blocks = (M // 64) * (N // 128)
    print(f"M={M:6d} blocks={blocks:5d} ({blocks / 256:5.2f}/CU)  {t:7.1f} us   {t / max(blocks / 256, 1):6.2f} us per block-per-CU   {6.0 * M * N * K / t / 1e6:7.1f} TF issued")
