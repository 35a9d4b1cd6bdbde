"""(ticks = 10 ns: the 100 MHz wall clock)  Workgroup timeline of one 3x3 launch on the 256-row tile kernel, per XCD (debug stamps: entry, first tile landed, K loop end,
staging barrier, -, exit).  Answers: how many workgroups are resident per XCD on average (32 CUs, one workgroup each), how long a
CU sits between the exit of one workgroup and the entry of the next.
usage: stamp_occupancy.py frames H C"""
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
torch.cuda.synchronize()
s, e = torch.cuda.Event(True), torch.cuda.Event(True)
s.record(); run(); e.record(); torch.cuda.synchronize()
plain_us = s.elapsed_time(e) * 1e3
nblk = 8 * 70000
buf = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
_hip.lib().td_debug_set_stamp_buffer(buf.data_ptr())
s.record(); run(); e.record()
torch.cuda.synchronize()
_hip.lib().td_debug_set_stamp_buffer(None)
us = s.elapsed_time(e) * 1e3
st = buf.view(nblk, 8).cpu()
idx = torch.nonzero(st[:, 0] > 0).flatten()
print(f"launch {plain_us:.1f} us plain, {us:.1f} us stamped; workgroups {idx.numel()}")
for xcd in range(8):
    sel = idx[(idx % 8) == xcd]
    t0, t5 = st[sel, 6].double(), st[sel, 7].double()  # wall clock (100 MHz)
    base = t0.min()
    span = (t5.max() - base).item()
    life = (t5 - t0)
    # events: +1 at entry, -1 at exit -> resident count over time
    ev = torch.cat([torch.stack([t0 - base, torch.ones_like(t0)], 1), torch.stack([t5 - base, -torch.ones_like(t5)], 1)])
    ev = ev[ev[:, 0].argsort()]
    res = ev[:, 1].cumsum(0)
    dt = ev[1:, 0] - ev[:-1, 0]
    avg = (res[:-1] * dt).sum().item() / span
    # gap seen by a "slot": k-th exit to (32 + k)-th entry
    ent, ext = t0.sort().values, t5.sort().values
    n = min(ext.numel(), ent.numel() - 32)
    gaps = (ent[32:32 + n] - ext[:n]) if n > 0 else torch.zeros(1)
    print(f"xcd {xcd}: {sel.numel():4d} wgs, span {span:9.0f} ticks ({span / us:7.1f} ticks/us), life median {life.median().item():8.0f}, avg resident {avg:5.1f} / 32, "
          f"max resident {res.max().item():.0f}, exit->next entry median {gaps.median().item():7.0f} p90 {gaps.quantile(0.9).item():7.0f}, first entries spread {(ent[min(31, ent.numel() - 1)] - ent[0]).item():7.0f}")
