"""GEMM microbenchmark on the shapes of the step (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aread_amd import _lib as L

SHAPES = [  # name, M, N, K, G, a_kc, b_kc
    ("fwd  L1 9728x1024x288", 9728, 1024, 288, 1, 1, 1),
    ("fwd  L2 9728x128x256 G4", 9728, 128, 256, 4, 1, 1),
    ("fwd  L3 9728x64x128 G4", 9728, 64, 128, 4, 1, 1),
    ("dgrad L1 9728x288x1024", 9728, 288, 1024, 1, 1, 0),
    ("dgrad L2 9728x256x128 G4", 9728, 256, 128, 4, 1, 0),
    ("wgrad L1 1024x288x9728", 1024, 288, 9728, 1, 0, 0),
    ("tower 9728x16x16 G12", 9728, 16, 16, 12, 1, 1),
]


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    for name, M, N, K, G, akc, bkc in SHAPES:
        A = torch.randn(G * M * K, device="cuda")
        B = torch.randn(G * N * K, device="cuda")
        C = torch.empty(G * M * N, device="cuda")
        lda, a_gs = (K, M * K) if akc else (M, K * M)
        ldb, b_gs = (K, N * K) if bkc else (N, K * N)
        fn = lambda: L.check(L.lib().aread_gemm(L.ptr(A), lda, a_gs, akc, L.ptr(B), ldb, b_gs, bkc, L.ptr(C), N, M * N, None, 0,
                                                M, N, K, G, 0, L.stream()))
        t = timeit(fn)
        fl = 2.0 * M * N * K * G
        print(f"{name:28s} {t:8.1f} us  {fl / t / 1e6:7.1f} TFLOP/s  ({fl / t / 1e6 / 157.3 * 100:4.1f}% of f32 MFMA peak)")
        if akc and bkc:
            f3 = lambda: L.check(L.lib().aread_gemm_bf16x3(L.ptr(A), lda, a_gs, L.ptr(B), ldb, b_gs, L.ptr(C), N, M * N, None, 0,
                                                           M, N, K, G, 0, L.stream()))
            t3 = timeit(f3)
            print(f"{'   (split-bf16, 3 MFMA products)':28s} {t3:8.1f} us  {fl / t3 / 1e6:7.1f} TFLOP/s algorithmic")
        if akc and bkc and G == 1:
            a2, b2 = A.view(M, K), B.view(N, K)
            t2 = timeit(lambda: torch.mm(a2, b2.t()))
            print(f"{'   (torch.mm / hipBLASLt f32)':28s} {t2:8.1f} us  {fl / t2 / 1e6:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
