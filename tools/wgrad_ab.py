"""Time the batched weight gradients of the trunk's trainable layers (layer2..4 of ResNet-101 at res 352) on synthetic
gradients; A/B an env knob by running twice (TD_WGRAD_WIDE=0 vs 1).  usage: wgrad_ab.py [slow frames, default 100]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tubedetr_amd import ops
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
LAYERS = [(44, 128, 512, 4), (22, 256, 1024, 23), (11, 512, 2048, 3)]  # (H, mid, out, blocks)
jobs, flops = [], 0.0
def t(*shape): return torch.relu(torch.randn(*shape, device=dev, generator=g)).bfloat16()
bufs = {}
for H, mid, out, nb in LAYERS:
    x_in = bufs.setdefault((H, out), t(frames, H, H, out)); x_mid = bufs.setdefault((H, mid), t(frames, H, H, mid))
    g_mid = bufs.setdefault(("g", H, mid), t(frames, H, H, mid)); g_out = bufs.setdefault(("g", H, out), t(frames, H, H, out))
    for b in range(nb):
        jobs.append((g_mid, x_in, 1, 1, 1, 0, None, out)); jobs.append((g_mid, x_mid, 3, 3, 1, 1, None, mid)); jobs.append((g_out, x_mid, 1, 1, 1, 0, None, mid))
        flops += 2.0 * frames * H * H * (mid * out * 2 + mid * mid * 9)
for _ in range(2): ops.conv_wgrad_batch(jobs)
torch.cuda.synchronize()
s, e = torch.cuda.Event(True), torch.cuda.Event(True)
s.record()
for _ in range(5): ops.conv_wgrad_batch(jobs)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 5
print(f"{len(jobs)} jobs, {flops / 1e12:.2f} TFLOP: {ms:.3f} ms  {flops / ms / 1e9:.1f} TF/s")
