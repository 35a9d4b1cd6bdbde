import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tubedetr_amd import ops
dev = torch.device("cuda:0")
n = 60500 * 1024
a = torch.randn(n, device=dev).bfloat16(); b = torch.randn(n, device=dev).bfloat16()
def timeit(f, reps=10):
    for _ in range(3): f()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True); s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / reps * 1e3
t = timeit(lambda: ops.add(a, b)); print(f"td_add bf16 (2R+1W, {3*n*2/1e6:.0f} MB): {t:.1f} us -> {3*n*2/t/1e6:.2f} TB/s")
t = timeit(lambda: torch.add(a, b)); print(f"torch.add bf16: {t:.1f} us -> {3*n*2/t/1e6:.2f} TB/s")
c = torch.empty_like(a)
t = timeit(lambda: c.copy_(a)); print(f"torch copy bf16 (1R+1W): {t:.1f} us -> {2*n*2/t/1e6:.2f} TB/s")
af = torch.randn(n // 2, device=dev); cf = torch.empty_like(af)
t = timeit(lambda: cf.copy_(af)); print(f"torch copy f32 (1R+1W, {2*af.numel()*4/1e6:.0f} MB): {t:.1f} us -> {2*af.numel()*4/t/1e6:.2f} TB/s")
big = torch.randn(512 * 1024 * 1024 // 4, device=dev); bo = torch.empty_like(big)
t = timeit(lambda: bo.copy_(big)); print(f"torch copy f32 512MB: {t:.1f} us -> {2*big.numel()*4/t/1e6:.2f} TB/s")
t = timeit(lambda: big.sum()); print(f"torch sum f32 512MB (read only): {t:.1f} us -> {big.numel()*4/t/1e6:.2f} TB/s")
t = timeit(lambda: bo.fill_(1.0)); print(f"torch fill f32 512MB (write only): {t:.1f} us -> {big.numel()*4/t/1e6:.2f} TB/s")
huge = torch.empty(4 * 1024**3 // 2, dtype=torch.bfloat16, device=dev); ho = torch.empty_like(huge)
t = timeit(lambda: ho.copy_(huge), 5); print(f"torch copy bf16 4GB: {t:.1f} us -> {2*huge.numel()*2/t/1e6:.2f} TB/s")
