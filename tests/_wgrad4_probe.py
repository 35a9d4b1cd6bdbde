"""Helper of test_bench_shapes_gpu.py::test_four_and_eight_wavefront_weight_gradient_tiles_equal_the_sixteen_wavefront_ones: a few wide weight-gradient
jobs (3x3 and pointwise, a k tile that ends inside the matrix, ragged row counts) in deterministic mode (one work item per output tile: a
fixed summation order), one SHA-256 per result.  The test runs it with TD_WGRAD_WIDE4=1, TD_WGRAD_WIDE8=1 and neither (the knobs are read once per process)."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tubedetr_amd  # noqa: E402
from tubedetr_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(29)
tubedetr_amd.set_deterministic(True)


def rnd(shape, scale=1.0, relu=False):
    x = torch.randn(shape, generator=g, device=dev) * scale
    return (x.relu() if relu else x).to(torch.bfloat16)


N = 37
jobs = []
# (Ci, H, Co, R, stride, pad): layer3 conv2 / conv3 / conv1, layer2 conv2 (K = 1152: the last 256-wide k tile is half empty), layer4 conv1
for Ci, H, Co, R, st, pad in [(256, 22, 256, 3, 1, 1), (256, 22, 1024, 1, 1, 0), (1024, 22, 256, 1, 1, 0), (128, 44, 256, 3, 1, 1), (2048, 11, 512, 1, 1, 0)]:
    Ho = (H + 2 * pad - R) // st + 1
    x = rnd((N, H, H, Ci), relu=True)
    gy = rnd((N, Ho, Ho, Co), 0.05)
    scale = torch.rand(Co, generator=g, device=dev) + 0.5
    jobs.append((gy, x, R, R, st, pad, scale, Ci))
outs = ops.conv_wgrad_batch(jobs)
torch.cuda.synchronize()
for o in outs:
    print(tuple(o.shape), hashlib.sha256(o.contiguous().view(torch.int32).cpu().numpy().tobytes()).hexdigest())
