"""Time ghf_input_proj_fwd at C3's shape (1 M rows, 128 -> 128), with and without the fused fp16 pieces."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from graph_hypernetwork_forge_amd import _native
dev = torch.device("cuda:0")
N, F, d = 1_000_000, 128, 128
x = torch.randn(N, F, device=dev); W = torch.randn(d, F, device=dev) * 0.1; b = torch.randn(d, device=dev)
hs = _native.alloc_split(N, d, _native.WLAYOUT_SPLIT2H, dev)
for split in (False, True):
    for _ in range(3):
        _native.input_proj_fwd(x, W, b, h_split=hs if split else None, split_layout=_native.WLAYOUT_SPLIT2H if split else 0)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(20):
        _native.input_proj_fwd(x, W, b, h_split=hs if split else None, split_layout=_native.WLAYOUT_SPLIT2H if split else 0)
    ev[1].record(); torch.cuda.synchronize()
    print(f"split={split}: {ev[0].elapsed_time(ev[1]) / 20:.3f} ms")
