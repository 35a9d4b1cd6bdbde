"""Per-workgroup cycle anatomy of one conv_gemm launch (debug stamps): prologue / K loop / epilogue phases."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tubedetr_amd import ops, _hip
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
res = len(sys.argv) > 4 and sys.argv[4] == "1"
dev = torch.device("cuda:0")
x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
b = torch.randn(N, device=dev); r = torch.randn(M, N, device=dev).bfloat16() if res else None
y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for _ in range(3): ops.linear_fwd(x, w, b, residual=r, relu=True, out=y)
nblk = 8 * 70000
buf = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
_hip.lib().td_debug_set_stamp_buffer(buf.data_ptr())
ops.linear_fwd(x, w, b, residual=r, relu=True, out=y)
torch.cuda.synchronize()
_hip.lib().td_debug_set_stamp_buffer(None)
st = buf.view(nblk, 8).cpu()
st = st[st[:, 0] > 0].double()
print("workgroups", st.shape[0])
d = st[:, 1:6] - st[:, 0:5]
names = ["setup+first tile landed", "K loop", "epilogue operand fetch issue + barrier", "LDS transpose", "epilogue math+stores"]
for i, n in enumerate(names): print(f"{n:40s} median {d[:, i].median().item():9.0f}  mean {d[:, i].mean().item():9.0f} cycles")
tot = st[:, 5] - st[:, 0]
print(f"{'total per workgroup':40s} median {tot.median().item():9.0f}  mean {tot.mean().item():9.0f}")
span = st[:, 5].max() - st[:, 0].min()
print("kernel span cycles", span.item(), " sum(wg cycles)/span =", (tot.sum() / span).item(), "avg resident workgroups")
