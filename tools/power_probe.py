"""Sustained clock / power while ONE kernel shape runs back to back (the 3x3 of layer3 on 256-row tiles, or the persistent 1x1).
usage: power_probe.py [3x3|1x1|mix|wgrad|l1|chain|pair] [seconds]   - prints rocm-smi power / sclk samples taken while the loop runs.
(wgrad = the trunk's 90 batched weight-gradient jobs at 400 slow frames, l1 = the fused 256-channel layer1 bottleneck at 1 000 frames,
chain = td_pw_chain2 on 800 layer3 frames, pair = the two launches it replaces on the same tensors; TD_HIP_LIB=<TD_CHAIN_ABL build> for ablations)"""
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
def make_wgrad():
    LAYERS = [(44, 128, 512, 4), (22, 256, 1024, 23), (11, 512, 2048, 3)]
    jobs, bufs = [], {}
    tt = lambda *shape: torch.relu(torch.randn(*shape, device=dev, generator=g)).bfloat16()
    for Hh, mid, out, nb in LAYERS:
        x_in = bufs.setdefault((Hh, out), tt(400, Hh, Hh, out)); x_mid = bufs.setdefault((Hh, mid), tt(400, Hh, Hh, mid))
        g_mid = bufs.setdefault(("g", Hh, mid), tt(400, Hh, Hh, mid)); g_out = bufs.setdefault(("g", Hh, out), tt(400, Hh, Hh, out))
        for _ in range(nb):
            jobs.append((g_mid, x_in, 1, 1, 1, 0, None, out)); jobs.append((g_mid, x_mid, 3, 3, 1, 1, None, mid)); jobs.append((g_out, x_mid, 1, 1, 1, 0, None, mid))
    return lambda: ops.conv_wgrad_batch(jobs)


def make_l1():
    from tubedetr_amd import _hip
    N, Hh = 1000, 88
    xin = torch.randn(N, Hh, Hh, 256, device=dev, generator=g).relu().bfloat16()
    out = torch.empty_like(xin)
    w1 = (torch.randn(64, 256, device=dev, generator=g) * 0.05).bfloat16(); w2 = (torch.randn(64, 576, device=dev, generator=g) * 0.05).bfloat16()
    w3 = (torch.randn(256, 64, device=dev, generator=g) * 0.05).bfloat16()
    b1, b2, b3 = (torch.randn(n_, device=dev, generator=g) for n_ in (64, 64, 256))
    L = _hip.lib()
    return lambda: _hip.check(L.td_bottleneck_fused(xin.data_ptr(), out.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), w3.data_ptr(), b3.data_ptr(),
                                                    None, None, N, Hh, Hh, 256, _hip.TD_BF16, _hip.stream_ptr()), "td_bottleneck_fused")


def make_chain(fused):
    M = frames * 484
    w1 = (torch.randn(256, 1024, device=dev, generator=g) * 0.02).bfloat16()
    b1 = torch.randn(256, device=dev, generator=g)
    y2 = torch.randn(M, 256, device=dev, generator=g).relu().bfloat16()
    h1 = torch.empty(M, 256, device=dev, dtype=torch.bfloat16)
    if fused:
        return lambda: ops.pw_chain2(y2, w3, b3, r3, w1, b1, out=y3, h1=h1)
    return lambda: (ops.linear_fwd(y2, w3, b3, residual=r3, relu=True, out=y3), ops.linear_fwd(y3, w1, b1, relu=True, out=h1))


fn = {"chain": lambda: make_chain(True), "pair": lambda: make_chain(False), "3x3": lambda: f3, "1x1": lambda: f1, "mix": lambda: (lambda: (f3(), f1())), "wgrad": make_wgrad, "l1": make_l1}[what]()
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
    reps = 5 if what in ("wgrad", "l1") else 50
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    per.append(e0.elapsed_time(e1) / reps * 1e3); n += reps
stop = True; th.join()
print(what, "us per call over time:", " ".join(f"{p:.0f}" for p in per[:: max(1, len(per) // 24)]))
for t_, s_ in samples[:: max(1, len(samples) // 6)]:
    print(f"t+{t_ - t0:5.1f}s {s_}")
