"""Per-workgroup cycle anatomy of one conv_wgrad launch.  usage: stamp_wgrad.py frames H C_in C_out R"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tubedetr_amd import ops, _hip
Nf, H, Ci, Co, R = (int(v) for v in sys.argv[1:6])
dev = torch.device("cuda:0")
x = torch.randn(Nf, H, H, Ci, device=dev).bfloat16()
g = torch.randn(Nf, H, H, Co, device=dev).bfloat16()
dw = torch.zeros(Co, R * R * Ci, device=dev)
for _ in range(3): ops.conv_wgrad(g, x, R, R, 1, R // 2, out=dw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.conv_wgrad(g, x, R, R, 1, R // 2, out=dw)
e1.record(); torch.cuda.synchronize()
print(f"launch avg {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
nblk = 4096
buf = torch.zeros(nblk * 40, dtype=torch.int64, device=dev)
_hip.lib().td_debug_set_stamp_buffer(buf.data_ptr())
ops.conv_wgrad(g, x, R, R, 1, R // 2, out=dw)
torch.cuda.synchronize()
_hip.lib().td_debug_set_stamp_buffer(None)
st = buf.view(nblk, 40).cpu()
st = st[st[:, 0] > 0].double()
print("workgroups", st.shape[0])
print(f"setup            median {(st[:,1]-st[:,0]).median().item():8.0f}")
print(f"loop (all)       median {(st[:,2]-st[:,1]).median().item():8.0f}")
print(f"atomics          median {(st[:,3]-st[:,2]).median().item():8.0f}")
print(f"first stage done median {(st[:,4]-st[:,1]).median().item():8.0f}")
per = []
for j in range(1, 32):
    ok = st[:, 4 + j] > 0
    if ok.sum() == 0: break
    per.append((st[ok, 4 + j] - st[ok, 3 + j]).median().item())
print("per-stage medians:", " ".join(f"{v:.0f}" for v in per))
span = st[:, 3].max() - st[:, 0].min()
print("kernel span cycles", span.item(), "start skew (max-min of start)", (st[:, 0].max() - st[:, 0].min()).item())
