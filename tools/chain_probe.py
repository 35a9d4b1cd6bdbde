"""The chained conv3 -> next-block conv1 kernel (td_pw_chain2, chain.hip) against the unfused pair on one layer3 shape:
bit-identity of both outputs (also at ragged row counts) and time per launch of either form in a sustained loop.
usage: chain_probe.py [frames] [reps]      (TD_HIP_LIB=<other build> for an A/B)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tubedetr_amd import ops

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1600
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
P = 256
torch.manual_seed(0)
w3 = (torch.randn(4 * P, P, device=dev) * (2.0 / P) ** 0.5).bfloat16()
w1 = (torch.randn(P, 4 * P, device=dev) * (0.5 / P) ** 0.5).bfloat16()
b3 = torch.randn(4 * P, device=dev) * 0.1
b1 = torch.randn(P, device=dev) * 0.1


def unfused(y2, res):
    out = ops.linear_fwd(y2, w3, b3, residual=res, relu=True)
    h1 = ops.linear_fwd(out, w1, b1, relu=True)
    return out, h1


ok = True
for M in (1, 31, 128, 129, 1000, 4097, 484 * 50 + 17):
    y2 = torch.randn(M, P, device=dev).relu().bfloat16()
    res = torch.randn(M, 4 * P, device=dev).relu().bfloat16()
    o_ref, h_ref = unfused(y2, res)
    guard_o = torch.full((M + 64, 4 * P), 7.0, device=dev, dtype=torch.bfloat16)
    guard_h = torch.full((M + 64, P), 7.0, device=dev, dtype=torch.bfloat16)
    o, h = ops.pw_chain2(y2, w3, b3, res, w1, b1, out=guard_o[:M], h1=guard_h[:M])
    torch.cuda.synchronize()
    eq_o, eq_h = torch.equal(o, o_ref), torch.equal(h, h_ref)
    clean = bool((guard_o[M:] == 7.0).all() and (guard_h[M:] == 7.0).all())
    d_o = (o.float() - o_ref.float()).abs().max().item()
    d_h = (h.float() - h_ref.float()).abs().max().item()
    print(f"M={M}: out identical {eq_o} (max diff {d_o:.3g}), h1 identical {eq_h} (max diff {d_h:.3g}), rows past M untouched {clean}", flush=True)
    ok = ok and eq_o and eq_h and clean

M = frames * 484
y2 = torch.randn(M, P, device=dev).relu().bfloat16()
res = torch.randn(M, 4 * P, device=dev).relu().bfloat16()
out = torch.empty(M, 4 * P, device=dev, dtype=torch.bfloat16)
h1 = torch.empty(M, P, device=dev, dtype=torch.bfloat16)
o_ref, h_ref = unfused(y2, res)
ops.pw_chain2(y2, w3, b3, res, w1, b1, out=out, h1=h1)
torch.cuda.synchronize()
print(f"M={M}: out identical {torch.equal(out, o_ref)}, h1 identical {torch.equal(h1, h_ref)}", flush=True)
ok = ok and torch.equal(out, o_ref) and torch.equal(h1, h_ref)
del o_ref, h_ref


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


o2 = torch.empty_like(out)
t_a = timeit(lambda: ops.linear_fwd(y2, w3, b3, residual=res, relu=True, out=o2))
t_b = timeit(lambda: ops.linear_fwd(o2, w1, b1, relu=True, out=h1))
t_c = timeit(lambda: ops.pw_chain2(y2, w3, b3, res, w1, b1, out=out, h1=h1))
by_pair = M * (256 + 1024 + 1024 + 1024 + 256) * 2
by_chain = M * (256 + 1024 + 1024 + 256) * 2
fl = 4.0 * M * 1024 * 256
print(f"rows {M}: conv3 {t_a:.1f} us + conv1 {t_b:.1f} us = {t_a + t_b:.1f} us ({by_pair / (t_a + t_b) / 1e6:.2f} TB/s of the pair's bytes);  "
      f"chain {t_c:.1f} us ({by_chain / t_c / 1e6:.2f} TB/s of its bytes, {fl / t_c / 1e6:.0f} TFLOP/s, {t_c * 1e3 / M:.3f} ns/row)  ratio {t_c / (t_a + t_b):.3f}")
print("PARITY", "OK" if ok else "FAILED")
