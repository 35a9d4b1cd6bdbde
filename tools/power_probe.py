"""Sustained clock / power while ONE kernel shape runs back to back (the 3x3 of layer3 on 256-row tiles, or the persistent 1x1).
usage: power_probe.py [3x3|1x1|mix] [seconds]   - prints rocm-smi power / sclk samples taken while the loop runs."""
import os, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tubedetr_amd import ops
what = sys.argv[1] if len(sys.argv) > 1 else "3x3"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
dev = torch.device("cuda:0")
frames = 800
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(frames, 22, 22, 256, device=dev, generator=g).bfloat16()
wf = (torch.randn(256, 9 * 256, device=dev, generator=g) * 0.02).bfloat16()
bias = torch.randn(256, device=dev, generator=g)
y = torch.empty(frames, 22, 22, 256, device=dev, dtype=torch.bfloat16)
w3 = (torch.randn(1024, 256, device=dev, generator=g) * 0.02).bfloat16()
b3 = torch.randn(1024, device=dev, generator=g)
r3 = torch.randn(frames * 484, 1024, device=dev, generator=g).bfloat16()
y3 = torch.empty(frames * 484, 1024, device=dev, dtype=torch.bfloat16)
f3 = lambda: ops.conv_fwd(x, wf, bias, 3, 3, 1, 1, relu=True, out=y)
f1 = lambda: ops.linear_fwd(y.view(-1, 256), w3, b3, residual=r3, relu=True, out=y3)
fn = {"3x3": f3, "1x1": f1, "mix": lambda: (f3(), f1())}[what]
samples = []
stop = False
def sampler():
    while not stop:
        try:
            out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
            samples.append((time.time(), out.strip()[:600]))
        except Exception as e:
            samples.append((time.time(), repr(e)))
        time.sleep(0.5)
for _ in range(5): fn()
torch.cuda.synchronize()
th = threading.Thread(target=sampler); th.start()
t0 = time.time(); n = 0
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
per = []
while time.time() - t0 < secs:
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    per.append(e0.elapsed_time(e1) / 50 * 1e3); n += 50
stop = True; th.join()
print(what, "us per call over time:", " ".join(f"{p:.0f}" for p in per[:: max(1, len(per) // 24)]))
for t_, s_ in samples[:: max(1, len(samples) // 6)]:
    print(f"t+{t_ - t0:5.1f}s {s_}")
