"""Per-kernel timing of the pieces of the path (development aid; bench.py is the judged benchmark)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import aread_amd
from aread_amd import _lib as L
from oracle import aread_oracle as O
from tests.test_gpu_embed import synth_x


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
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    spec = O.amazon_spec()
    rng = np.random.default_rng(0)
    mh = {"multi_hot_flag": list(spec.multi_hot_flag), "itemid_idx": 0, "seq_maxlen": 5, "method": "mean"}
    emb = aread_amd.FeaturesEmbedding(list(spec.field_dims), 32, mh).cuda()
    for B in (8192, 65536, 262144):
        x = synth_x(spec, rng, B)
        x[:, 2] = rng.integers(0, 25, B)
        xd = torch.from_numpy(x).cuda()
        w = emb.embedding_dict.weight.detach()
        out = torch.empty((B, 9, 32), device="cuda")
        off = emb._offsets_dev(xd.device)
        f = lambda: L.check(L.lib().aread_embed_fwd(L.ptr(xd), B, 17, L.ptr(off), L.ptr(w), w.shape[0], 32, 7, 2, 5, 2,
                                                    None, B, L.ptr(out), None, L.stream()))
        t = timeit(f)
        print(f"B={B}: embed_fwd {t:8.1f} us   alg {B*3396/t/1e6:7.1f} GB/s  read-stream {B*2244/t/1e6:7.1f} GB/s")
        t = timeit(lambda: aread_amd.RowPlan(xd, 2, 25))
        print(f"B={B}: plan_build {t:8.1f} us")
        grad = torch.zeros_like(w)
        dout = torch.randn((B, 9, 32), device="cuda")
        t = timeit(lambda: emb.scatter_grad(xd, dout, grad), iters=20)
        print(f"B={B}: embed_bwd (keys+sort+reduce) {t:8.1f} us   alg {B*5572/t/1e6:7.1f} GB/s")
    n = w.numel()
    grad = torch.empty_like(w)
    part = torch.empty(L.lib().aread_l2_partials(), device="cuda")
    t = timeit(lambda: L.check(L.lib().aread_l2_table(L.ptr(w), n, 1e-5, 1.0, L.ptr(grad), L.ptr(part), L.stream())))
    print(f"l2_table n={n}: {t:8.1f} us   {2*n*4/t/1e6:7.1f} GB/s")
    t = timeit(lambda: L.check(L.lib().aread_l2_table(L.ptr(w), n, 1e-5, 1.0, None, L.ptr(part), L.stream())))
    print(f"l2_table loss-only: {t:8.1f} us   {n*4/t/1e6:7.1f} GB/s")
    t = timeit(lambda: grad.copy_(w))
    print(f"torch copy (ref): {t:8.1f} us   {2*n*4/t/1e6:7.1f} GB/s")


if __name__ == "__main__":
    main()
