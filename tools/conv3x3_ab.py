"""Time the spatial (3x3) trunk convolutions, forward and input-gradient, on the benchmark's shapes.  A/B an env knob of the
library by running it twice, e.g. TD_CONV_TAP_UNIFORM=0 vs 1 (the knobs are read once per process).
usage: conv3x3_ab.py [frames]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tubedetr_amd import ops

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 125
dev = torch.device("cuda:0")
SHAPES = [  # (H=W, C, stride): layer3 / layer2 / layer4 conv2 and the strided first blocks
    (22, 256, 1), (44, 128, 1), (11, 512, 1), (44, 256, 2), (88, 128, 2), (22, 512, 2)]
g = torch.Generator(device=dev).manual_seed(0)


def timeit(fn, reps=20):
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


for H, Cc, st in SHAPES:
    x = torch.randn(frames, H, H, Cc, device=dev, generator=g).bfloat16()
    wf = (torch.randn(Cc, 9 * Cc, device=dev, generator=g) * 0.02).bfloat16()
    bias = torch.randn(Cc, device=dev, generator=g)
    Ho = (H + 2 - 3) // st + 1
    y = torch.empty(frames, Ho, Ho, Cc, device=dev, dtype=torch.bfloat16)
    us = timeit(lambda: ops.conv_fwd(x, wf, bias, 3, 3, st, 1, relu=True, out=y))
    fl = 2.0 * frames * Ho * Ho * Cc * 9 * Cc
    print(f"fwd   N={frames} {H}x{H} C={Cc} s{st}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s")
    if st == 2 and H % 2 == 0:  # input gradient of the stride-2 layer (TD_DGRAD_S2_PARITY: four parity-class launches vs one 9-tap gather)
        nb = max(1, frames // 5)    # backward runs on the slow frames only
        gy = torch.randn(nb, Ho, Ho, Cc, device=dev, generator=g).bfloat16()
        wd = (torch.randn(Cc, 9 * Cc, device=dev, generator=g) * 0.02).bfloat16()
        act = torch.randn(nb, H, H, Cc, device=dev, generator=g).relu().bfloat16()
        dx = torch.empty(nb, H, H, Cc, device=dev, dtype=torch.bfloat16)
        us = timeit(lambda: ops.conv_dgrad(gy, wd, (H, H), 3, 3, 2, 1, mask_src=act, out=dx))
        print(f"dgrad N={nb} {H}x{H} C={Cc} s2: {us:8.1f} us  {2.0 * nb * Ho * Ho * Cc * 9 * Cc / us / 1e6:7.1f} TF/s (nominal 9-tap FLOPs)")
    if st == 1:
        gy = torch.randn(frames, Ho, Ho, Cc, device=dev, generator=g).bfloat16()
        wd = (torch.randn(Cc, 9 * Cc, device=dev, generator=g) * 0.02).bfloat16()
        dx = torch.empty(frames, H, H, Cc, device=dev, dtype=torch.bfloat16)
        us = timeit(lambda: ops.conv_dgrad(gy, wd, (H, H), 3, 3, st, 1, mask_src=x, out=dx))
        print(f"dgrad N={frames} {H}x{H} C={Cc} s{st}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s")

# pointwise layers with K >= 512 (conv1 of layer2..4 forward, conv3 input gradient): rows x K -> N
for rows, K, Nn in [(frames * 484, 1024, 256), (frames * 121, 2048, 512), (frames * 1936, 512, 128), (frames * 484, 512, 256)]:
    x = torch.randn(rows, K, device=dev, generator=g).bfloat16()
    w = (torch.randn(Nn, K, device=dev, generator=g) * 0.02).bfloat16()
    b = torch.randn(Nn, device=dev, generator=g)
    y = torch.empty(rows, Nn, device=dev, dtype=torch.bfloat16)
    us = timeit(lambda: ops.linear_fwd(x, w, b, relu=True, out=y))
    by = (rows * K + Nn * K + rows * Nn) * 2
    print(f"1x1   M={rows} K={K} N={Nn}: {us:8.1f} us  {2.0 * rows * K * Nn / us / 1e6:7.1f} TF/s  {by / us / 1e3:7.1f} GB/s")

# persistent 1x1 instance (K <= 256): conv3 of a layer3 bottleneck with its residual, and the layer1 shape
for rows, K, Nn in [(frames * 484, 256, 1024), (frames * 7744, 64, 256)]:
    x = torch.randn(rows, K, device=dev, generator=g).bfloat16()
    w = (torch.randn(Nn, K, device=dev, generator=g) * 0.02).bfloat16()
    b = torch.randn(Nn, device=dev, generator=g)
    r = torch.randn(rows, Nn, device=dev, generator=g).bfloat16()
    y = torch.empty(rows, Nn, device=dev, dtype=torch.bfloat16)
    us = timeit(lambda: ops.linear_fwd(x, w, b, residual=r, relu=True, out=y))
    by = (rows * K + Nn * K + 2 * rows * Nn) * 2
    print(f"1x1+res M={rows} K={K} N={Nn}: {us:8.1f} us  {by / us / 1e3:7.1f} GB/s")
