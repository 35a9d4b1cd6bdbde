"""Time the attention core on the three problem shapes of the headline clip (bf16)."""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tubedetr_amd import ops

dev = torch.device("cuda:0")
for name, (B, H, Lq, Lk, need_w) in {"encoder": (25, 8, 151, 151, False), "temporal": (1, 8, 100, 100, False), "cross": (100, 8, 1, 151, True)}.items():
    E = H * 32
    q = torch.randn(B, Lq, E, device=dev).bfloat16(); k = torch.randn(B, Lk, E, device=dev).bfloat16(); v = torch.randn(B, Lk, E, device=dev).bfloat16()
    do = torch.randn(B, Lq, E, device=dev).bfloat16(); dw = torch.randn(B, Lq, Lk, device=dev) if need_w else None
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    def fwd():
        return ops.mha_fwd(q, k, v, None, H, 1 / math.sqrt(32), need_wavg=need_w, dropout_p=0.1, seed=3)
    out, probs, _ = fwd()
    def bwd():
        ops.mha_bwd(q, k, v, do, probs, dw, H, 1 / math.sqrt(32), dq, dk, dv, dropout_p=0.1, seed=3)
    for fn, nm in ((fwd, "fwd"), (bwd, "bwd")):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"{name:9s} {nm}: {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us (incl. output allocation / launch gaps)")
