"""Time td_cross_q1_dmem (and the frame-core kernels) at the bench shape: 1 600 frames x 151 rows, six layers."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tubedetr_amd import ops
F, S, H, E, NL = 1600, 151, 8, 256, 6
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.3).bfloat16()
coef = r(F * S, 96)
layers = [(r(F, H * E), r(F, H * E + H)) for _ in range(NL)]
mem, pos = r(F * S, E), r(F * S, E)
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
print("dmem %.1f us" % timeit(lambda: ops.cross_q1_dmem(coef, layers, F, S, H, E)))
u = layers[0][0]
probs, _, zext = ops.cross_q1_fwd(u, mem, pos, None, F, S, H, need_wavg=True, dropout_p=0.1, seed=3)
print("fwd %.1f us" % timeit(lambda: ops.cross_q1_fwd(u, mem, pos, None, F, S, H, need_wavg=True, dropout_p=0.1, seed=3)))
dz = layers[0][1]
print("bwd (coef) %.1f us" % timeit(lambda: ops.cross_q1_bwd_coef(u, mem, pos, probs, dz, None, coef, 0, F, S, H, dropout_p=0.1, seed=3)))
