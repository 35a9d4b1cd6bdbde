"""Run a single conv_gemm shape repeatedly (for rocprofv3 --pmc runs).  usage: one_conv.py M N K [residual] [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tubedetr_amd import ops
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
res = len(sys.argv) > 4 and sys.argv[4] == "1"
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dev = torch.device("cuda:0")
x = torch.randn(M, K, device=dev).bfloat16()
w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
b = torch.randn(N, device=dev)
r = torch.randn(M, N, device=dev).bfloat16() if res else None
y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for _ in range(reps):
    ops.linear_fwd(x, w, b, residual=r, relu=True, out=y)
torch.cuda.synchronize()
s, e = torch.cuda.Event(True), torch.cuda.Event(True)
s.record()
for _ in range(reps):
    ops.linear_fwd(x, w, b, residual=r, relu=True, out=y)
e.record(); torch.cuda.synchronize()
print(f"M={M} N={N} K={K} res={res}: {s.elapsed_time(e)/reps*1e3:.1f} us")
