#!/usr/bin/env python
"""Write / copy bandwidth of plain device kernels at the sizes of the step's activations (the roof of the GEMM epilogues)."""
import torch


def t(fn, iters=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for rows, cols in [(9728, 1024), (9728, 288), (9728, 512), (9728 * 4, 1024)]:
    x = torch.empty((rows, cols), device="cuda")
    y = torch.empty((rows, cols), device="cuda")
    mb = rows * cols * 4 / 1e6
    tf = t(lambda: x.fill_(1.0))
    tc = t(lambda: y.copy_(x))
    tr = t(lambda: x.sum())
    print(f"{rows}x{cols} fp32 ({mb:.1f} MB): fill {tf:6.2f} us ({mb / tf:.2f} TB/s)  copy {tc:6.2f} us ({2 * mb / tc:.2f} TB/s)  "
          f"sum {tr:6.2f} us ({mb / tr:.2f} TB/s read)")
