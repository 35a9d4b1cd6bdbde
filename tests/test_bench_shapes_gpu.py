"""The bf16-only kernel instances that carry the benchmarked step, at EXACTLY the shapes bench.py launches them with
(8 clips of cfg3 per step = a 1 000-frame trunk forward, a 200-frame trunk backward), each against a plain PyTorch fp32
evaluation of the same op on the same bf16-rounded operands:

  conv_gemm_big8_kernel<true, *>    layer3 3x3        M = 484 000, N = 256, K = 2304      forward and input gradient
  conv_gemm_big8_kernel<false, *>   layer3 conv1      M = 484 000, N = 256, K = 1024
  pw_resident2_kernel               layer3 conv3      M = 484 000, N = 1024, K = 256      + residual + ReLU
  pw_resident2_kernel               layer1 conv3      M = 7 744 000, N = 256, K = 64      + residual + ReLU
  conv_gemm_big8(n)_kernel / conv_gemm_kernel (tap-uniform)  the stride-2 layers of a stage's first block (3x3 and the 1x1 downsample)
  pw_chain2_kernel                  layer3 conv3 + identity chained with the next conv1   M = 484 000 / 774 400, 256 -> 1024 -> 256
  conv_wgrad_wide_batch_kernel      the trunk's batched weight-gradient table at 200 slow frames (one job per layer shape)

The forward-type results are compared on sampled row ranges (first / middle / last rows of the launch: tile 0, an interior
tile, the ragged last tile) - the reference of a whole 484 000 x 2304 launch would be another GEMM library's result, not
a check; weight gradients are full reductions and are compared whole, against fp32 matmuls over the same rows.
Tolerance: operands are identical bf16 values on both sides and both accumulate in fp32, so only the output rounding (bf16: at most
2^-8 of the value) and the fp32 summation order differ - PER ELEMENT |err| <= 2^-8 |ref| + 2e-3 max |ref| (round 6; it used to be one
1.2e-2 max |ref| for the whole tensor, three times what those two effects explain)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

FRAMES_FWD, FRAMES_BWD = 1000, 200  # 8 clips x (100 fast + 25 slow) frames forward, 8 x 25 slow frames backward


def dev():
    return torch.device("cuda:0")


def assert_bf16(got, ref, what=None):
    """every element within the bf16 output rounding of the fp32 reference plus a summation-order allowance"""
    got, ref = got.detach().double(), ref.detach().double()
    err = (got - ref).abs()
    bound = ref.abs() * 2.0**-8 + 2e-3 * ref.abs().max()
    bad = err > bound
    assert not bool(bad.any()), (what, int(bad.sum()), float((err / bound).max()))


def rel_err(got, ref):
    got, ref = got.detach().double(), ref.detach().double()
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


def _rand(shape, g, scale=1.0, relu=False):
    x = torch.randn(shape, generator=g, device=dev(), dtype=torch.float32) * scale
    if relu:
        x = x.relu()  # trunk activations are post-ReLU (half zeros): what the kernels see in the network
    return x.to(torch.bfloat16)


def _frames_sample(n):
    return sorted({0, 1, n // 2, n - 2, n - 1})


def test_layer3_conv3x3_forward_and_dgrad_at_bench_shape():
    from tubedetr_amd import ops

    g = torch.Generator(device=dev()).manual_seed(3)
    N, H, W, C = FRAMES_FWD, 22, 22, 256
    x = _rand((N, H, W, C), g, relu=True)
    w = (torch.randn(C, C, 3, 3, generator=g, device=dev()) / math.sqrt(9 * C)).to(torch.bfloat16).float()
    bias = torch.randn(C, generator=g, device=dev())
    wf, wd, b_out, _ = ops.weight_prep(w, torch.bfloat16, bias=bias)
    y = ops.conv_fwd(x, wf, b_out, 3, 3, 1, 1, relu=True)
    assert y.shape == (N, H, W, C)
    fr = _frames_sample(N)
    xs = x[fr].float().permute(0, 3, 1, 2)
    ref = F.relu(F.conv2d(xs, w, bias, padding=1)).permute(0, 2, 3, 1)
    assert_bf16(y[fr].float(), ref)
    # input gradient at the backward's shape (200 frames: 379 row tiles of 256 on 256 CUs), ReLU mask of the producing layer fused
    Nb = FRAMES_BWD
    gy = _rand((Nb, H, W, C), g)
    act = _rand((Nb, H, W, C), g, relu=True)
    dx = ops.conv_dgrad(gy, wd, (H, W), 3, 3, 1, 1, mask_src=act)
    fr = _frames_sample(Nb)
    gs = gy[fr].float().permute(0, 3, 1, 2)
    ref = F.conv_transpose2d(gs, w, padding=1).permute(0, 2, 3, 1) * (act[fr].float() > 0)
    assert_bf16(dx[fr].float(), ref)


@pytest.mark.parametrize("shape", [("layer3.0 downsample (strided 1x1 on the 256-row tiles)", 44, 512, 1024, 1), ("layer2.0 downsample (strided 1x1, short K)", 88, 256, 512, 1),
                                   ("layer3.0 conv2 (strided 3x3 on the 256-row tiles)", 44, 256, 256, 3)])
def test_stage_entry_strided_layers_at_bench_shape(shape):
    from tubedetr_amd import ops

    _, HW, C, Nc, ksz = shape
    g = torch.Generator(device=dev()).manual_seed(11)
    N = FRAMES_FWD
    x = _rand((N, HW, HW, C), g, relu=True)
    w = (torch.randn(Nc, C, ksz, ksz, generator=g, device=dev()) / math.sqrt(ksz * ksz * C)).to(torch.bfloat16).float()
    bias = torch.randn(Nc, generator=g, device=dev())
    wf, _, b_out, _ = ops.weight_prep(w, torch.bfloat16, bias=bias)
    y = ops.conv_fwd(x, wf, b_out, ksz, ksz, 2, ksz // 2, relu=ksz == 3)
    assert y.shape == (N, HW // 2, HW // 2, Nc)
    fr = _frames_sample(N)
    ref = F.conv2d(x[fr].float().permute(0, 3, 1, 2), w, bias, stride=2, padding=ksz // 2).permute(0, 2, 3, 1)
    if ksz == 3:
        ref = ref.relu()
    assert_bf16(y[fr].float(), ref, shape[0])


@pytest.mark.parametrize("shape", [("layer3.conv1 (256-row tiles)", 484000, 1024, 256, False), ("layer3.conv3 (persistent)", 484000, 256, 1024, True),
                                   ("layer1.conv3 (persistent)", 7744000, 64, 256, True), ("layer2.conv1 (256-row tiles, 128 wide)", 1936000, 512, 128, False)])
def test_pointwise_layers_at_bench_shape(shape):
    from tubedetr_amd import ops

    _, M, K, Nc, with_res = shape
    g = torch.Generator(device=dev()).manual_seed(5)
    x = _rand((M, K), g, relu=True)
    w = (torch.randn(Nc, K, generator=g, device=dev()) / math.sqrt(K)).to(torch.bfloat16)
    bias = torch.randn(Nc, generator=g, device=dev())
    res = _rand((M, Nc), g, relu=True) if with_res else None
    y = ops.linear_fwd(x, w, bias, residual=res, relu=True)
    for a, b in ((0, 4096), (M // 2 - 1000, M // 2 + 3000), (M - 4096, M)):
        ref = x[a:b].float() @ w.float().t() + bias
        if res is not None:
            ref = ref + res[a:b].float()
        assert_bf16(y[a:b].float(), ref.relu(), (shape[0], a))


@pytest.mark.parametrize("frames", [FRAMES_FWD, 1600, 7], ids=["484000_rows", "774400_rows", "ragged_3388_rows"])
def test_chained_conv3_conv1_pair_at_bench_shape(frames):
    """td_pw_chain2 (chain.hip): conv3 + identity + ReLU of a layer3 block and conv1 + ReLU of the next in one launch, 484 000 x 256 -> 1024 -> 256.
    (a) bit-identical to the two launches it replaces (pw_resident2_kernel, conv_gemm_big8_kernel<false>); (b) both results against fp32 torch on
    the same bf16 operands, per element: |err| <= 2^-8 |ref| + 2e-3 max|ref| (output rounding + fp32 summation order; the second product sees the
    bf16-rounded first result on both sides); (c) rows past M stay untouched."""
    from tubedetr_amd import ops

    P, M = 256, frames * 484
    g = torch.Generator(device=dev()).manual_seed(11)
    y2 = _rand((M, P), g, relu=True)
    res = _rand((M, 4 * P), g, relu=True)
    w3 = (torch.randn(4 * P, P, generator=g, device=dev()) / math.sqrt(P)).to(torch.bfloat16)
    w1 = (torch.randn(P, 4 * P, generator=g, device=dev()) / math.sqrt(4 * P)).to(torch.bfloat16)
    b3 = torch.randn(4 * P, generator=g, device=dev()) * 0.1
    b1 = torch.randn(P, generator=g, device=dev()) * 0.1
    guard_o = torch.full((M + 130, 4 * P), 3.0, device=dev(), dtype=torch.bfloat16)
    guard_h = torch.full((M + 130, P), 3.0, device=dev(), dtype=torch.bfloat16)
    out, h1 = ops.pw_chain2(y2, w3, b3, res, w1, b1, out=guard_o[:M], h1=guard_h[:M])
    out_u = ops.linear_fwd(y2, w3, b3, residual=res, relu=True)
    h1_u = ops.linear_fwd(out_u, w1, b1, relu=True)
    assert torch.equal(out, out_u) and torch.equal(h1, h1_u), "the chained launch differs from the two launches it replaces"
    assert bool((guard_o[M:] == 3.0).all()) and bool((guard_h[M:] == 3.0).all()), "rows past M were written"
    for a, b in ((0, min(M, 4096)), (max(0, M // 2 - 1000), min(M, M // 2 + 3000)), (max(0, M - 4096), M)):
        ref_o = (y2[a:b].float() @ w3.float().t() + b3 + res[a:b].float()).relu()
        err = (out[a:b].float() - ref_o).abs()
        assert bool((err <= ref_o.abs() * 2.0**-8 + 2e-3 * ref_o.abs().max()).all()), (frames, a, "out")
        ref_h = (out[a:b].float() @ w1.float().t() + b1).relu()
        err = (h1[a:b].float() - ref_h).abs()
        assert bool((err <= ref_h.abs() * 2.0**-8 + 2e-3 * ref_h.abs().max()).all()), (frames, a, "h1")


def test_persistent_256_row_kernel_equals_one_tile_per_workgroup():
    """The persistent form of conv_gemm_big8_kernel (next tile's prologue pieces issued under this tile's epilogue, hand-counted vmcnt / lgkmcnt
    waits) against its one-tile-per-workgroup form (TD_CONV_BIG_PERSIST=0, the documented fallback), bit for bit: plain, residual and residual +
    ReLU-mask epilogues, row counts that end in a ragged tile.  A compiler that reschedules a scalar load into those counted sequences would
    show up here as a changed bit."""
    import os
    import subprocess
    import sys

    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_persist_probe.py")
    outs = []
    for knob in ("1", "0"):
        env = dict(os.environ, TD_CONV_BIG_PERSIST=knob)
        r = subprocess.run([sys.executable, probe], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if len(ln.split()) >= 2 and len(ln.split()[-1]) == 64]
        assert len(lines) == 5, r.stdout
        outs.append(lines)
    assert outs[0] == outs[1], "\n".join(f"{a}  |  {b}" for a, b in zip(*outs))


def test_four_and_eight_wavefront_weight_gradient_tiles_equal_the_sixteen_wavefront_ones():
    """conv_wgrad_wide4_batch_kernel (256 x 256 tiles on four wavefronts with 128 x 128 wave tiles, hand-allocated accumulator file, counted vmcnt on a
    four-stage ring; TD_WGRAD_WIDE4=1) and conv_wgrad_wide8_batch_kernel (eight wavefronts, 128 x 64 wave tiles; TD_WGRAD_WIDE8=1) against
    conv_wgrad_wide_batch_kernel (sixteen wavefronts, the default) in deterministic mode - one work item per output tile, the same summation order
    per element in all three: bit for bit."""
    import os
    import subprocess
    import sys

    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_wgrad4_probe.py")
    outs = []
    for knobs in ({"TD_WGRAD_WIDE4": "1"}, {"TD_WGRAD_WIDE8": "1"}, {}):
        env = dict(os.environ, TD_WGRAD_WIDE4="0", TD_WGRAD_WIDE8="0")
        env.update(knobs)
        r = subprocess.run([sys.executable, probe], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if len(ln.split()) >= 2 and len(ln.split()[-1]) == 64]
        assert len(lines) == 5, r.stdout
        outs.append(lines)
    assert outs[0] == outs[2], "\n".join(f"{a}  |  {b}" for a, b in zip(outs[0], outs[2]))
    assert outs[1] == outs[2], "\n".join(f"{a}  |  {b}" for a, b in zip(outs[1], outs[2]))


def _wgrad_ref(gy, x, R, stride, pad):
    """dW [Co, Ci, R, R] = sum over output pixels of gy^T x(shifted): fp32 matmuls over the same rows (NHWC operands)."""
    N, H, W, Ci = x.shape
    _, Ho, Wo, Co = gy.shape
    G = gy.float().reshape(-1, Co)
    xp = F.pad(x.float(), (0, 0, pad, pad, pad, pad))
    dW = torch.empty((Co, Ci, R, R), dtype=torch.float32, device=x.device)
    for r in range(R):
        for s in range(R):
            xs = xp[:, r : r + (Ho - 1) * stride + 1 : stride, s : s + (Wo - 1) * stride + 1 : stride, :].reshape(-1, Ci)
            dW[:, :, r, s] = G.t() @ xs
    return dW


def test_trunk_weight_gradient_table_at_bench_shape():
    """One job per distinct layer shape of the trunk's backward at 200 slow frames of res 352 (layer2: 44 x 44, layer3: 22 x 22,
    layer4: 11 x 11), all in ONE td_conv_wgrad_batch launch like td_resnet_bwd issues it (wide tiles, split and unsplit work
    items, FrozenBN scale folded, parameter-layout output)."""
    from tubedetr_amd import ops

    g = torch.Generator(device=dev()).manual_seed(9)
    Nb = FRAMES_BWD
    # (Ci, H, W, Co, R, stride, pad)
    shapes = [(256, 88, 88, 128, 1, 1, 0), (128, 88, 88, 128, 3, 2, 1), (128, 44, 44, 512, 1, 1, 0), (256, 88, 88, 512, 1, 2, 0),
              (512, 44, 44, 128, 1, 1, 0), (128, 44, 44, 128, 3, 1, 1),
              (512, 44, 44, 256, 1, 1, 0), (256, 44, 44, 256, 3, 2, 1), (256, 22, 22, 1024, 1, 1, 0), (512, 44, 44, 1024, 1, 2, 0),
              (1024, 22, 22, 256, 1, 1, 0), (256, 22, 22, 256, 3, 1, 1),
              (1024, 22, 22, 512, 1, 1, 0), (512, 22, 22, 512, 3, 2, 1), (512, 11, 11, 2048, 1, 1, 0), (1024, 22, 22, 2048, 1, 2, 0),
              (2048, 11, 11, 512, 1, 1, 0), (512, 11, 11, 512, 3, 1, 1)]
    jobs, keep = [], []
    for i, (Ci, H, W, Co, R, st, pad) in enumerate(shapes):
        Ho, Wo = (H + 2 * pad - R) // st + 1, (W + 2 * pad - R) // st + 1
        x = _rand((Nb, H, W, Ci), g, relu=True)
        gy = _rand((Nb, Ho, Wo, Co), g, scale=0.05)
        scale = torch.rand(Co, generator=g, device=dev()) + 0.5
        jobs.append((gy, x, R, R, st, pad, scale, Ci))
        keep.append((gy, x, R, st, pad, scale))
    outs = ops.conv_wgrad_batch(jobs)
    torch.cuda.synchronize()
    for got, (gy, x, R, st, pad, scale), shp in zip(outs, keep, shapes):
        ref = _wgrad_ref(gy, x, R, st, pad) * scale.view(-1, 1, 1, 1)
        assert got.shape == ref.shape
        assert rel_err(got, ref) < 2e-3, shp  # fp32 results of identical bf16 operands: only the summation order differs


def test_layer4_layers_on_the_256_row_kernel_at_bench_shape():
    """The members of conv_gemm_big8_kernel that layer3's shapes do not reach (round 5's persistent form walks them differently): two
    column tiles per row tile (layer4's 3x3: 512 output channels), the residual instantiation (layer4's conv3: K = 512 -> 2048 with the
    identity added in the MFMA layout), and residual + ReLU mask together (the input gradient of layer4's conv1: 512 -> 2048 channels)."""
    from tubedetr_amd import ops

    g = torch.Generator(device=dev()).manual_seed(19)
    N, H, W, C = FRAMES_FWD, 11, 11, 512
    # 3x3, 512 -> 512: forward (bias + ReLU) and input gradient (ReLU mask)
    x = _rand((N, H, W, C), g, relu=True)
    w = (torch.randn(C, C, 3, 3, generator=g, device=dev()) / math.sqrt(9 * C)).to(torch.bfloat16).float()
    bias = torch.randn(C, generator=g, device=dev())
    wf, wd, b_out, _ = ops.weight_prep(w, torch.bfloat16, bias=bias)
    y = ops.conv_fwd(x, wf, b_out, 3, 3, 1, 1, relu=True)
    fr = _frames_sample(N)
    ref = F.relu(F.conv2d(x[fr].float().permute(0, 3, 1, 2), w, bias, padding=1)).permute(0, 2, 3, 1)
    assert_bf16(y[fr].float(), ref)
    gy = _rand((N, H, W, C), g)
    act = _rand((N, H, W, C), g, relu=True)
    dx = ops.conv_dgrad(gy, wd, (H, W), 3, 3, 1, 1, mask_src=act)
    ref = F.conv_transpose2d(gy[fr].float().permute(0, 3, 1, 2), w, padding=1).permute(0, 2, 3, 1) * (act[fr].float() > 0)
    assert_bf16(dx[fr].float(), ref)
    # conv3: rows x 512 -> 2048 with residual + ReLU (pointwise K >= 512: the 256-row kernel's residual instantiation)
    M = N * H * W
    h2 = _rand((M, 512), g, relu=True)
    w3 = (torch.randn(2048, 512, generator=g, device=dev()) / math.sqrt(512)).to(torch.bfloat16)
    b3 = torch.randn(2048, generator=g, device=dev())
    res = _rand((M, 2048), g, relu=True)
    out = ops.linear_fwd(h2, w3, b3, residual=res, relu=True)
    for a, b in ((0, 4096), (M // 2 - 1000, M // 2 + 3000), (M - 4096, M)):
        ref = (h2[a:b].float() @ w3.float().t() + b3 + res[a:b].float()).relu()
        assert_bf16(out[a:b].float(), ref, a)
    # input gradient of conv1 (2048 -> 512 forward): g [M, 512] x W [512, 2048] + residual (the identity branch's gradient), masked by the block input
    gh1 = _rand((M, 512), g)
    w1 = (torch.randn(512, 2048, generator=g, device=dev()) / math.sqrt(2048)).to(torch.bfloat16).float()
    _, w1d, _, _ = ops.weight_prep(w1.view(512, 2048, 1, 1), torch.bfloat16)
    gres = _rand((M, 2048), g)
    xin = _rand((M, 2048), g, relu=True)
    dxin = ops.conv_dgrad(gh1.view(N, H, W, 512), w1d, (H, W), 1, 1, 1, 0, residual=gres.view(N, H, W, 2048), mask_src=xin.view(N, H, W, 2048)).view(M, 2048)
    for a, b in ((0, 4096), (M // 2 - 1000, M // 2 + 3000), (M - 4096, M)):
        ref = (gh1[a:b].float() @ w1 + gres[a:b].float()) * (xin[a:b].float() > 0)
        assert_bf16(dxin[a:b].float(), ref, a)
