"""Times of _native.matmul_tn / matmul_nn at the weight generator's backward shapes (C3: R=64, T=64, Hh=128?, d=128)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graph_hypernetwork_forge_amd import _native
dev = torch.device("cuda:0")
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
R = 64
for Hh in (128, 256):
    for (K, M, N, what) in ((R, 16384, Hh, "tn dW_out"), (R, 128, Hh, "tn dW_out(bias head)"), (R, Hh, Hh, "tn dW_hidden"), (R, Hh, 64, "tn dW_in")):
        A, B = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
        print(f"Hh={Hh} {what:22s} A[{K},{M}]^T B[{K},{N}]: {t(lambda: _native.matmul_tn(A, B)):8.1f} us")
    for (M, K, N, what) in ((R, 16384, Hh, "nn dy@W_out"), (R, Hh, Hh, "nn dy@W_hidden"), (R, Hh, 64, "nn dy@W_in")):
        X, W = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev)
        print(f"Hh={Hh} {what:22s} X[{M},{K}] W[{K},{N}]: {t(lambda: _native.matmul_nn(X, W)):8.1f} us")
