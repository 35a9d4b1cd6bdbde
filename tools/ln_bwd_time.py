"""Time td_add_layernorm_bwd at the step's three shapes (bf16): encoder tokens 60 400 x 256, decoder rows 1 600 x 256, RoBERTa 480 x 768.
usage: python tools/ln_bwd_time.py   (prints us per launch, HIP-event timed over 200 back-to-back launches)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tubedetr_amd import ops

dev = torch.device("cuda:0")
for rows, cols, with_extra in ((60400, 256, True), (60400, 256, False), (1600, 256, True), (480, 768, True)):
    g = torch.Generator().manual_seed(1)
    dy = torch.randn(rows, cols, generator=g).to(dev, torch.bfloat16)
    s = torch.randn(rows, cols, generator=g).to(dev, torch.bfloat16)
    ex = torch.randn(rows, cols, generator=g).to(dev, torch.bfloat16) if with_extra else None
    gamma = torch.rand(cols, generator=g).to(dev) + 0.5
    mean, rstd = s.float().mean(1).contiguous(), (s.float().var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
    # the C entry point itself, outputs allocated once (the Python wrapper's three allocations cost more than the small launches)
    from tubedetr_amd import _hip
    ds, dg, db = torch.empty_like(dy), torch.zeros(cols, device=dev), torch.zeros(cols, device=dev)
    lib, p, st, dt = _hip.lib(), ops.ptr, _hip.stream_ptr(), _hip.dtype_code(dy.dtype)
    call = lambda: lib.td_add_layernorm_bwd(p(dy), p(s), p(mean), p(rstd), p(gamma), p(ex), p(ds), p(dg), p(db), rows, cols, dt, st)
    for _ in range(10):
        assert call() == 0
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(200):
        call()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1000 / 200
    mb = rows * cols * 2 * (4 if with_extra else 3) / 1e6
    print(f"{rows:6d} x {cols:4d} extra={with_extra}: {us:7.1f} us per launch = {mb / us / 1e3:5.2f} TB/s")
