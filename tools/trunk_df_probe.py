"""Depth-first no-grad trunk pass (resnet_exec.hip: run_depth_first) against the layer-by-layer pass: bit equality of the features
and time per pass for a list of TD_TRUNK_DF settings.  `python tools/trunk_df_probe.py [frames] [res]` (default 1600 352: the
no-grad pass of the benchmarked 16-clip step).  Also prints a device-to-device copy bandwidth ladder (buffers from 16 MB to 2 GB:
what the 256 MiB Infinity Cache does for a write-then-read-back pattern)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TD_ALLOW_RANDOM_TEXT_ENCODER", "1")

from tubedetr_amd.functional import invalidate_prepared  # noqa: E402
from tubedetr_amd.models.backbone import Backbone  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1600
RES = int(sys.argv[2]) if len(sys.argv) > 2 else 352
dev = torch.device("cuda:0")
torch.manual_seed(0)
bb = Backbone("resnet101", True, False, False).to(dev)
body = bb.body
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randint(0, 256, (N, 3, RES, RES), generator=g, device=dev, dtype=torch.uint8)


def run(setting, env=None, reps=3):
    for k in ("TD_TRUNK_DF", "TD_TRUNK_DF_INPLACE", "TD_TRUNK_DF_MB"):
        os.environ.pop(k, None)
    os.environ["TD_TRUNK_DF"] = setting
    for k, v in (env or {}).items():
        os.environ[k] = v
    with torch.no_grad():
        f = body(x, torch.bfloat16)  # warm-up (weights prepared once, workspace cached by the allocator)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        ev[0].record()
        for i in range(reps):
            f = body(x, torch.bfloat16)
            ev[i + 1].record()
        torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    return f, ms[len(ms) // 2]


ref, t0 = run("0")
print(f"frames {N} res {RES}: layer-by-layer {t0:.2f} ms")
cases = [("auto", None), ("auto", {"TD_TRUNK_DF_INPLACE": "0"}), ("auto", {"TD_TRUNK_DF_MB": "128"}), ("auto", {"TD_TRUNK_DF_MB": "256"}),
         ("auto", {"TD_TRUNK_DF_MB": "384"}), ("33,0,0", None), ("0,134,0", None), ("0,0,270", None), ("0,67,0", None), ("0,100,0", None),
         ("0,200,0", None), ("0,268,0", None), ("0,400,0", None), ("16,0,0", None), ("66,0,0", None), ("0,134,0", {"TD_TRUNK_DF_INPLACE": "0"}),
         ("0,0,0", None), ("0", None)]
for setting, env in cases:
    f, t = run(setting, env)
    same = torch.equal(f, ref)
    err = (f.float() - ref.float()).abs().max().item() / max(ref.float().abs().max().item(), 1e-9)
    print(f"TD_TRUNK_DF={setting:10s} {env or ''}: {t:8.2f} ms  ({t - t0:+.2f})  bit-identical={same} rel_err={err:.2e}", flush=True)

print("copy ladder (dst.copy_(src), bytes moved = 2 x size):")
for mb in (16, 32, 64, 96, 128, 192, 256, 512, 1024, 2048):
    n = mb * 1048576 // 2
    a = torch.empty(n, dtype=torch.bfloat16, device=dev).normal_()
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = max(4, 4096 // mb)
    e0.record()
    for i in range(reps):
        (b if i % 2 == 0 else a).copy_(a if i % 2 == 0 else b)  # ping-pong: every pass reads what the previous one wrote
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"  {mb:5d} MB: {2 * mb / 1024 / (ms * 1e-3) / 1e3 * 1.073741824:7.2f} TB/s", flush=True)
    del a, b
