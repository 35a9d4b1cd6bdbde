"""Helper of test_bench_shapes_gpu.py::test_persistent_256_row_kernel_equals_one_tile_per_workgroup: runs the members of
conv_gemm_big8_kernel on seeded operands (plain, residual, residual + ReLU mask epilogues; a ragged last row tile) and prints one
SHA-256 per result.  The test runs it twice, with TD_CONV_BIG_PERSIST=1 and =0 (the knob is read once per process)."""
import hashlib
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tubedetr_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(23)


def rnd(shape, scale=1.0, relu=False):
    x = torch.randn(shape, generator=g, device=dev) * scale
    return (x.relu() if relu else x).to(torch.bfloat16)


def digest(t):
    return hashlib.sha256(t.contiguous().view(torch.int16).cpu().numpy().tobytes()).hexdigest()


N = 301  # 301 x 22 x 22 = 145 684 rows: not a multiple of 256 (ragged last tile), more tiles than CUs
x = rnd((N, 22, 22, 256), relu=True)
w = (torch.randn(256, 9 * 256, generator=g, device=dev) / math.sqrt(2304)).to(torch.bfloat16)
b = torch.randn(256, generator=g, device=dev)
print("3x3 forward        ", digest(ops.conv_fwd(x, w, b, 3, 3, 1, 1, relu=True)))
gy = rnd((N, 22, 22, 256), 0.05)
print("3x3 dgrad + mask   ", digest(ops.conv_dgrad(gy, w, (22, 22), 3, 3, 1, 1, mask_src=x)))
M = N * 484
x1 = rnd((M, 1024), relu=True)
w1 = (torch.randn(256, 1024, generator=g, device=dev) / 32).to(torch.bfloat16)
print("1x1 K=1024 forward ", digest(ops.linear_fwd(x1, w1, b, relu=True)))
M4 = 1001 * 121  # layer4 rows, ragged
x4 = rnd((M4, 512), relu=True)
w4 = (torch.randn(2048, 512, generator=g, device=dev) / math.sqrt(512)).to(torch.bfloat16)
b4 = torch.randn(2048, generator=g, device=dev)
r4 = rnd((M4, 2048), relu=True)
print("K=512 + residual   ", digest(ops.linear_fwd(x4, w4, b4, residual=r4, relu=True)))
m4 = rnd((M4, 2048), relu=True)
print("K=512 + res + mask ", digest(ops.linear_fwd(x4, w4, None, residual=r4, mask_src=m4)))
