"""Per-workgroup cycle anatomy of one 3x3 (or 1x1) launch on the 256-row tile instance (debug stamps of conv_gemm_big_kernel).
usage: stamp_big.py frames H C [pw K N]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tubedetr_amd import ops, _hip
frames, H, Cc = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
x = torch.randn(frames, H, H, Cc, device=dev).bfloat16()
wf = (torch.randn(Cc, 9 * Cc, device=dev) * 0.02).bfloat16()
bias = torch.randn(Cc, device=dev)
y = torch.empty(frames, H, H, Cc, device=dev, dtype=torch.bfloat16)
run = lambda: ops.conv_fwd(x, wf, bias, 3, 3, 1, 1, relu=True, out=y)
for _ in range(3): run()
nblk = 8 * 70000
buf = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
_hip.lib().td_debug_set_stamp_buffer(buf.data_ptr())
s, e = torch.cuda.Event(True), torch.cuda.Event(True)
s.record(); run(); e.record()
torch.cuda.synchronize()
_hip.lib().td_debug_set_stamp_buffer(None)
st = buf.view(nblk, 8).cpu()
st = st[st[:, 0] > 0].double()
nk = 9 * Cc // 64
print("workgroups", st.shape[0], "k tiles", nk, "launch us", s.elapsed_time(e) * 1e3)
ph = {"first tile landed": (0, 1), "K loop": (1, 2), "operand fetch + barrier": (2, 3), "epilogue": (3, 5)}
for n, (a, b) in ph.items():
    d = st[:, b] - st[:, a]
    print(f"{n:28s} median {d.median().item():9.0f}  mean {d.mean().item():9.0f}  p90 {d.quantile(0.9).item():9.0f}")
loop = st[:, 2] - st[:, 1]
print(f"cycles per k tile: median {loop.median().item() / nk:7.0f}   (MFMA floor 2 waves x 64 MFMA x 16.5 = 2112)")
tot = st[:, 5] - st[:, 0]
span = st[:, 5].max() - st[:, 0].min()
print(f"total per workgroup median {tot.median().item():9.0f}; kernel span {span.item():.0f} ticks; avg resident workgroups {(tot.sum() / span).item():.1f}; ticks per us {span.item() / (s.elapsed_time(e) * 1e3):.1f}")
