"""Time td_pw_chain2 alone on the layer3 shape (for A/B builds of chain.hip: TD_HIP_LIB=...).  usage: chain_time.py [frames] [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tubedetr_amd import ops
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1600
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
P, M = 256, frames * 484
torch.manual_seed(0)
w3 = (torch.randn(4 * P, P, device=dev) * (2.0 / P) ** 0.5).bfloat16()
w1 = (torch.randn(P, 4 * P, device=dev) * (0.5 / P) ** 0.5).bfloat16()
b3 = torch.randn(4 * P, device=dev) * 0.1
b1 = torch.randn(P, device=dev) * 0.1
y2 = torch.randn(M, P, device=dev).relu().bfloat16()
res = torch.randn(M, 4 * P, device=dev).relu().bfloat16()
out = torch.empty(M, 4 * P, device=dev, dtype=torch.bfloat16)
h1 = torch.empty(M, P, device=dev, dtype=torch.bfloat16)
for _ in range(3):
    ops.pw_chain2(y2, w3, b3, res, w1, b1, out=out, h1=h1)
torch.cuda.synchronize()
s, e = torch.cuda.Event(True), torch.cuda.Event(True)
s.record()
for _ in range(reps):
    ops.pw_chain2(y2, w3, b3, res, w1, b1, out=out, h1=h1)
e.record()
torch.cuda.synchronize()
t = s.elapsed_time(e) / reps * 1e3
print(f"{os.environ.get('TD_HIP_LIB', 'default').split('_')[-1]}: rows {M}: chain {t:.1f} us, {t * 1e3 / M:.3f} ns/row, {M * 2560 * 2 / t / 1e6:.2f} TB/s, {4.0 * M * 1024 * 256 / t / 1e6:.0f} TFLOP/s")
