"""GPU: the training side's fp32-MFMA tile GEMM (ph_gemm32) alone: correctness against torch and time per launch vs shape"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyphonicformer_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def run(M, N, K, kcA, kcB, ksplit=1, bias=False, iters=50):
    A = torch.randn((M, K) if kcA else (K, M), generator=g).to(dev)
    B = torch.randn((N, K) if kcB else (K, N), generator=g).to(dev)
    bs = torch.randn(N, generator=g).to(dev) if bias else None
    C = torch.empty((ksplit, M, N), device=dev)
    call = lambda: _lib.check(lib.ph_gemm32(_lib.ptr(A), A.shape[1], kcA, _lib.ptr(B), B.shape[1], kcB, _lib.ptr(C), N, M, N, K, ksplit,
                                            _lib.ptr(bs), _lib.stream_ptr()), "ph_gemm32")
    call()
    ref = (A if kcA else A.t()).double() @ (B.t() if kcB else B).double()
    if bias:
        ref = ref + bs.double()
    err = float((C.sum(0).double() - ref).abs().max() / ref.abs().max())
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        call()
    e.record()
    torch.cuda.synchronize()
    return err, s.elapsed_time(e) / iters * 1e3


for (M, N, K, kcA, kcB, ks, b) in [(306, 256, 32, 1, 1, 1, False), (306, 256, 256, 1, 1, 1, False), (306, 256, 256, 1, 1, 1, True), (306, 256, 256, 1, 0, 1, False),
                                   (306, 2048, 256, 1, 1, 1, True), (306, 256, 2048, 1, 1, 8, False), (306, 256, 2048, 1, 1, 1, False),
                                   (2048, 256, 306, 0, 0, 1, False), (256, 256, 306, 0, 0, 1, False), (64, 64, 256, 1, 1, 1, False), (306, 19, 256, 1, 1, 1, True)]:
    err, us = run(M, N, K, kcA, kcB, ks, b)
    print(f"M {M} N {N} K {K} kcA {kcA} kcB {kcB} ksplit {ks} bias {b}: rel err {err:.1e}, {us:.1f} us per launch (back to back)")
