"""Per-kernel parity on a real MI355X: every C-ABI op against a plain PyTorch fp32 (CPU) reference of the
same op.  fp32 mode must agree to fp32 round-off; bf16 mode is compared against the reference evaluated on
bf16-rounded inputs with a bf16-sized tolerance (stated per test)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.bfloat16]
TOL = {torch.float32: 2e-5, torch.bfloat16: 1.2e-2}  # max|err| / max|ref|


def dev():
    return torch.device("cuda:0")


def rel_err(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


def rnd(shape, g, dt, scale=1.0):
    x = torch.randn(shape, generator=g) * scale
    return x.to(dt).float()  # values exactly representable in dt


def nhwc(x, dt):  # NCHW cpu float -> NHWC device dt
    return x.permute(0, 2, 3, 1).contiguous().to(dev(), dt)


def from_nhwc(y):
    return y.float().cpu().permute(0, 3, 1, 2)


CONVS = [
    # N, Cin, H, W, Cout, R, stride, pad
    (2, 64, 12, 12, 64, 1, 1, 0),
    (3, 64, 11, 13, 256, 1, 1, 0),
    (2, 128, 14, 14, 128, 3, 1, 1),
    (2, 128, 15, 13, 128, 3, 2, 1),
    (2, 256, 9, 9, 512, 1, 2, 0),
    (2, 8, 30, 34, 64, 7, 2, 3),
    (1, 512, 6, 6, 2048, 1, 1, 0),
    (5, 256, 22, 22, 256, 3, 1, 1),
]


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("cfg", CONVS)
def test_conv_fwd_dgrad_wgrad(cfg, dt):
    from tubedetr_amd import ops

    N, Ci, H, W, Co, R, st, pad = cfg
    g = torch.Generator().manual_seed(1)
    x = rnd((N, Ci, H, W), g, dt)
    w = rnd((Co, Ci, R, R), g, dt, 1.0 / math.sqrt(Ci * R * R))
    bias = torch.randn(Co, generator=g)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, bias, stride=st, padding=pad)
    res = rnd(tuple(y_ref.shape), g, dt)
    out_ref = F.relu(y_ref + res)
    gy = rnd(tuple(y_ref.shape), g, dt)
    out_ref.backward(gy)
    gpre = (gy * (out_ref > 0)).detach()  # gradient at the conv output

    wf, wd, b_out, _ = ops.weight_prep(w.to(dev()), dt, bias=bias.to(dev()))
    xd = nhwc(x, dt)
    y = ops.conv_fwd(xd, wf, b_out, R, R, st, pad, residual=nhwc(res, dt), relu=True)
    assert rel_err(from_nhwc(y), out_ref) < TOL[dt]

    gd = nhwc(gpre, dt)
    dx = ops.conv_dgrad(gd, wd, (H, W), R, R, st, pad)
    assert rel_err(from_nhwc(dx), xr.grad) < TOL[dt]

    dwk = ops.conv_wgrad(gd, xd, R, R, st, pad)
    dW = ops.wgrad_finalize(dwk, None, (Co, Ci, R, R), Ci)
    assert rel_err(dW, wr.grad) < TOL[dt]


@pytest.mark.parametrize("dt", DT)
def test_conv_wgrad_batch_matches_autograd(dt):
    """All weight gradients of a mixed set of layers in one batched launch: parameter-layout output, FrozenBN scale
    folded, un-split jobs (plain stores) and split jobs (long M: atomics on a zeroed output) side by side, stale
    output buffers overwritten."""
    from tubedetr_amd import ops

    g = torch.Generator().manual_seed(21)
    cfgs = CONVS + [(44, 64, 28, 28, 64, 1, 1, 0), (60, 64, 24, 24, 128, 3, 1, 1)]  # the last two are long enough to split (> 512 stages of 64 rows)
    jobs, refs = [], []
    for i, (N, Ci, H, W, Co, R, st, pad) in enumerate(cfgs):
        x = rnd((N, Ci, H, W), g, dt)
        Ho, Wo = (H + 2 * pad - R) // st + 1, (W + 2 * pad - R) // st + 1
        gy = rnd((N, Co, Ho, Wo), g, dt)
        w = torch.zeros(Co, Ci, R, R, requires_grad=True)
        F.conv2d(x, w, None, stride=st, padding=pad).backward(gy)
        scale = (torch.rand(Co, generator=g) + 0.5) if i % 2 == 0 else None
        refs.append(w.grad * (scale.view(-1, 1, 1, 1) if scale is not None else 1.0))
        jobs.append((nhwc(gy, dt), nhwc(x, dt), R, R, st, pad, scale.to(dev()) if scale is not None else None, Ci))
    outs = ops.conv_wgrad_batch(jobs)
    for got, ref, cfg in zip(outs, refs, cfgs):
        assert got.shape == ref.shape
        assert rel_err(got, ref) < TOL[dt], cfg
    # padded source channels (the stem: 3 real channels stored as 8): only ci_real channels are written
    N, H, W, Co = 2, 30, 34, 64
    x = rnd((N, 3, H, W), g, dt)
    gy = rnd((N, Co, 15, 17), g, dt)
    w = torch.zeros(Co, 3, 7, 7, requires_grad=True)
    F.conv2d(x, w, None, stride=2, padding=3).backward(gy)
    xp = torch.zeros(N, 8, H, W)
    xp[:, :3] = x
    (dw,) = ops.conv_wgrad_batch([(nhwc(gy, dt), nhwc(xp, dt), 7, 7, 2, 3, None, 3)])
    assert rel_err(dw, w.grad) < TOL[dt]


def test_conv_wgrad_batch_wide_tiles():
    """bf16 jobs with >= 4096 reduction rows and 128-multiple channel counts take the sixteen-wavefront wide-tile instance
    (conv_wgrad_wide_batch_kernel): every tile class (256x256, 128x256, 256x128; 3x3 stride 1 / 2 and pointwise), split
    (atomics) and un-split jobs, partial last k tile (K = 1152), ragged last stage, FrozenBN scale folding - in one
    launch, next to a small job that stays on the 128x128 instance."""
    from tubedetr_amd import ops

    dt = torch.bfloat16
    g = torch.Generator().manual_seed(23)
    cfgs = [(8, 128, 28, 27, 128, 3, 1, 1), (52, 256, 26, 25, 256, 3, 1, 1), (5, 1024, 30, 31, 256, 1, 1, 0), (40, 512, 30, 30, 128, 1, 1, 0),
            (6, 128, 28, 28, 512, 1, 1, 0), (8, 128, 57, 57, 128, 3, 2, 1), (2, 64, 12, 12, 64, 1, 1, 0), (3, 512, 40, 40, 512, 3, 1, 1),
            (50, 128, 28, 27, 128, 3, 1, 1)]  # M = 33 800 / 36 000 / 37 800 rows: split in two (atomics), the others unsplit
    jobs, refs = [], []
    for i, (N, Ci, H, W, Co, R, st, pad) in enumerate(cfgs):
        x = rnd((N, Ci, H, W), g, dt).to(dev())
        Ho, Wo = (H + 2 * pad - R) // st + 1, (W + 2 * pad - R) // st + 1
        gy = rnd((N, Co, Ho, Wo), g, dt).to(dev())
        scale = (torch.rand(Co, generator=g) + 0.5).to(dev()) if i % 2 == 0 else None
        ref = torch.nn.grad.conv2d_weight(x, (Co, Ci, R, R), gy, stride=st, padding=pad)  # fp32 on the GPU, same bf16 values
        refs.append(ref * (scale.view(-1, 1, 1, 1) if scale is not None else 1.0))
        to_rows = lambda t_: t_.permute(0, 2, 3, 1).contiguous().to(dt)
        jobs.append((to_rows(gy), to_rows(x), R, R, st, pad, scale, Ci))
    outs = ops.conv_wgrad_batch(jobs)
    for got, ref, cfg in zip(outs, refs, cfgs):
        assert got.shape == ref.shape
        assert rel_err(got, ref) < TOL[dt], cfg
    outs2 = ops.conv_wgrad_batch(jobs)  # stale outputs are overwritten, split jobs re-zeroed
    for a, b in zip(outs, outs2):
        assert rel_err(a, b) < 1e-5


@pytest.mark.parametrize("dt", DT)
def test_anisotropic_stride_and_padding(dt):
    """Forward geometry with different vertical / horizontal stride and padding (td_conv_desc.aniso), generic kernel path."""
    from tubedetr_amd import ops

    g = torch.Generator().manual_seed(51)
    N, Ci, H, W, Co = 3, 8, 21, 17, 64
    x = rnd((N, Ci, H, W), g, dt)
    w = rnd((Co, Ci, 7, 4), g, dt, 0.1)
    bias = torch.randn(Co, generator=g)
    ref = F.relu(F.conv2d(x, w, bias, stride=(2, 1), padding=(3, 2)))
    Ho, Wo = ref.shape[2], ref.shape[3]
    wf = w.permute(0, 2, 3, 1).reshape(Co, -1).contiguous().to(dev(), dt)  # [Co][R][S][C], K contiguous
    y = torch.empty((N, Ho, Wo, Co), dtype=dt, device=dev())
    ops.conv_gemm_raw(nhwc(x, dt), wf, y, ops._desc(N, H, W, Ci, Ho, Wo, 7, 4, 2, 3, 0, Co, Co, stride_w=1, pad_w=2), ops._epi(bias.to(dev()), None, None, True))
    assert rel_err(from_nhwc(y), ref) < TOL[dt]


def test_stem_pixel_pair_form_equals_the_7x7_convolution():
    """td_resnet_fwd's stem_pairs layout: 4-channel bf16 pixels taken two at a time + td_stem_pair_weights + a 7x4 stride-(2,1)
    pad-(3,2) convolution == the 7x7 stride-2 pad-3 convolution of the 3-channel frames (bias + ReLU), incl. the image borders."""
    import ctypes as C
    from tubedetr_amd import _hip, ops

    dt = torch.bfloat16
    g = torch.Generator().manual_seed(53)
    N, H, W, Co = 3, 46, 60, 64
    x = rnd((N, 3, H, W), g, dt)
    w = rnd((Co, 3, 7, 7), g, dt, 0.1)
    bias = torch.randn(Co, generator=g)
    ref = F.relu(F.conv2d(x, w, bias, stride=2, padding=3))
    Ho, Wo = ref.shape[2], ref.shape[3]
    wf8, _, b_out, _ = ops.weight_prep(w.to(dev()), dt, bias=bias.to(dev()), need_dgrad=False, cpad=8)
    wp = torch.empty((Co, 7 * 4 * 8), dtype=dt, device=dev())
    _hip.check(_hip.lib().td_stem_pair_weights(wf8.data_ptr(), wp.data_ptr(), Co, _hip.TD_BF16, _hip.stream_ptr()), "td_stem_pair_weights")
    x4 = ops.nchw_to_nhwc(x.to(dev()), dt, 4)  # [N, H, W, 4] == [N, H, W/2, 8]
    y = torch.empty((N, Ho, Wo, Co), dtype=dt, device=dev())
    ops.conv_gemm_raw(x4.view(N, H, W // 2, 8), wp, y, ops._desc(N, H, W // 2, 8, Ho, Wo, 7, 4, 2, 3, 0, Co, Co, stride_w=1, pad_w=2), ops._epi(b_out, None, None, True))
    assert rel_err(from_nhwc(y), ref) < TOL[dt]
    y8 = ops.conv_fwd(ops.nchw_to_nhwc(x.to(dev()), dt, 8), wf8, b_out, 7, 7, 2, 3, relu=True)
    assert rel_err(y, y8) < 4e-3  # same products, different summation order


@pytest.mark.parametrize("size", [(3, 46, 60), (2, 352, 352), (5, 64, 96), (1, 30, 34), (2, 224, 224)])
def test_fused_stem_equals_conv_relu_maxpool(size):
    """td_stem_pool (conv7x7s2 + bias + ReLU + maxpool3x3s2 in one pass, the 64-channel map kept in LDS) against
    F.max_pool2d(F.relu(F.conv2d(...))) in fp32, and bit-for-bit against the two-launch pipeline it replaces (pixel-pair
    convolution written as bf16, then td_maxpool3x3s2): image borders, pooled sizes that do not fill the tiles, both tile widths."""
    from tubedetr_amd import _hip, ops

    dt = torch.bfloat16
    N, H, W = size
    Co = 64
    g = torch.Generator().manual_seed(H + W)
    x = rnd((N, 3, H, W), g, dt)
    w = rnd((Co, 3, 7, 7), g, dt, 0.1)
    bias = torch.randn(Co, generator=g)
    ref = F.max_pool2d(F.relu(F.conv2d(x, w, bias, stride=2, padding=3)), 3, 2, 1)
    PH, PW = ref.shape[2], ref.shape[3]
    wf8, _, b_out, _ = ops.weight_prep(w.to(dev()), dt, bias=bias.to(dev()), need_dgrad=False, cpad=8)
    wp = torch.empty((Co, 7 * 4 * 8), dtype=dt, device=dev())
    _hip.check(_hip.lib().td_stem_pair_weights(wf8.data_ptr(), wp.data_ptr(), Co, _hip.TD_BF16, _hip.stream_ptr()), "td_stem_pair_weights")
    x4 = ops.nchw_to_nhwc(x.to(dev()), dt, 4)  # [N, H, W, 4] == [N, H, W/2, 8]
    y = torch.full((N, PH, PW, Co), float("nan"), dtype=dt, device=dev())
    _hip.check(_hip.lib().td_stem_pool(x4.data_ptr(), wp.data_ptr(), b_out.data_ptr(), y.data_ptr(), N, H, W, _hip.TD_BF16, _hip.stream_ptr()), "td_stem_pool")
    assert rel_err(from_nhwc(y), ref) < TOL[dt]
    Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    c = torch.empty((N, Ho, Wo, Co), dtype=dt, device=dev())
    ops.conv_gemm_raw(x4.view(N, H, W // 2, 8), wp, c, ops._desc(N, H, W // 2, 8, Ho, Wo, 7, 4, 2, 3, 0, Co, Co, stride_w=1, pad_w=2), ops._epi(b_out, None, None, True))
    two = ops.maxpool3x3s2(c)
    # same bf16 products; the fp32 sums of the two kernels run in different orders, so a conv output may round to the neighbouring bf16
    assert rel_err(y, two) < 8e-3
    assert (y.float() - two.float()).abs().gt(0).float().mean().item() < 0.05


@pytest.mark.parametrize("cfg", [(2, 24, 32, 256), (3, 19, 27, 256), (2, 24, 32, 64), (1, 9, 50, 64), (2, 88, 88, 256), (1, 5, 3, 64),
                                 (4, 72, 70, 256), (8, 60, 90, 64)])  # (the last two: more tiles than CUs - the XCD-contiguous tile walk, uneven ranges, cut tiles)
def test_fused_frozen_bottleneck_equals_the_three_convolutions(cfg):
    """td_bottleneck_fused (a whole frozen layer1 block in one launch: conv1 -> conv2 3x3 -> conv3 (+ downsample) + identity + ReLU,
    the 64-channel tensors kept in LDS) against the block in fp32 torch on the same bf16 weights, and against the layer-by-layer
    kernels it replaces: tiles cut by the image border, maps smaller than a tile, both block types."""
    from tubedetr_amd import _hip, ops

    dt = torch.bfloat16
    N, H, W, Cin = cfg
    g = torch.Generator().manual_seed(H * W + Cin)
    x = rnd((N, Cin, H, W), g, dt).relu()
    w1 = rnd((64, Cin, 1, 1), g, dt, 1.0 / math.sqrt(Cin))
    w2 = rnd((64, 64, 3, 3), g, dt, 1.0 / math.sqrt(576))
    w3 = rnd((256, 64, 1, 1), g, dt, 1.0 / 8)
    wd = rnd((256, Cin, 1, 1), g, dt, 1.0 / 8) if Cin == 64 else None
    b1, b2, b3, bd = (torch.randn(n_, generator=g) * 0.3 for n_ in (64, 64, 256, 256))
    h = F.relu(F.conv2d(x, w1, b1)).to(dt).float()          # the fused kernel keeps the inner tensors as bf16, like the separate launches do
    h = F.relu(F.conv2d(h, w2, b2, padding=1)).to(dt).float()
    idt = x if wd is None else F.conv2d(x, wd, bd)
    ref = F.relu(F.conv2d(h, w3, b3) + idt)
    d = dev()
    prep = lambda w_, b_: ops.weight_prep(w_.to(d), dt, bias=b_.to(d), need_dgrad=False)
    (w1f, _, b1f, _), (w2f, _, b2f, _), (w3f, _, b3f, _) = prep(w1, b1), prep(w2, b2), prep(w3, b3)
    wdf, bdf = (None, None) if wd is None else prep(wd, bd)[0::2]
    xd = nhwc(x, dt)
    out = torch.full((N, H, W, 256), float("nan"), dtype=dt, device=d)
    _hip.check(_hip.lib().td_bottleneck_fused(xd.data_ptr(), out.data_ptr(), w1f.data_ptr(), b1f.data_ptr(), w2f.data_ptr(), b2f.data_ptr(), w3f.data_ptr(),
                                              b3f.data_ptr(), _hip.ptr(wdf), _hip.ptr(bdf), N, H, W, Cin, _hip.TD_BF16, _hip.stream_ptr()), "td_bottleneck_fused")
    assert torch.isfinite(out.float()).all()
    assert rel_err(from_nhwc(out), ref) < TOL[dt], cfg
    # the launches it replaces
    h1 = ops.conv_fwd(xd, w1f, b1f, 1, 1, 1, 0, relu=True)
    h2 = ops.conv_fwd(h1, w2f, b2f, 3, 3, 1, 1, relu=True)
    idn = xd if wd is None else ops.conv_fwd(xd, wdf, bdf, 1, 1, 1, 0)
    sep = ops.conv_fwd(h2, w3f, b3f, 1, 1, 1, 0, residual=idn, relu=True)
    assert rel_err(out, sep) < 8e-3, cfg


@pytest.mark.parametrize("dt", DT)
def test_frozen_bn_fold_and_mask_epilogues(dt):
    from tubedetr_amd import ops

    g = torch.Generator().manual_seed(2)
    N, Ci, H, W, Co = 2, 64, 10, 10, 128
    x = rnd((N, Ci, H, W), g, dt)
    w = rnd((Co, Ci, 3, 3), g, torch.float32, 0.05)
    bn = [torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g) * 0.1, torch.randn(Co, generator=g) * 0.1, torch.rand(Co, generator=g) + 0.5]
    scale = bn[0] * (bn[3] + 1e-5).rsqrt()
    shift = bn[1] - bn[2] * scale
    ref = F.relu(F.conv2d(x, w, padding=1) * scale[None, :, None, None] + shift[None, :, None, None])
    wf, wd, b_out, sc = ops.weight_prep(w.to(dev()), dt, bn=[b.to(dev()) for b in bn])
    assert rel_err(sc, scale) < 1e-5 and rel_err(b_out, shift) < 1e-5
    y = ops.conv_fwd(nhwc(x, dt), wf, b_out, 3, 3, 1, 1, relu=True)
    assert rel_err(from_nhwc(y), ref) < TOL[dt] * 2
    # dgrad with residual + mask epilogue: dx = (dgrad(g) + r) * (m > 0)
    gy = rnd((N, Co, H, W), g, dt)
    r = rnd((N, Ci, H, W), g, dt)
    m = rnd((N, Ci, H, W), g, dt)
    wfold = (w * scale[:, None, None, None])
    if dt == torch.bfloat16:
        wfold = wfold.bfloat16().float()
    dx_ref = (torch.nn.grad.conv2d_input((N, Ci, H, W), wfold, gy, padding=1) + r) * (m > 0)
    dx = ops.conv_dgrad(nhwc(gy, dt), wd, (H, W), 3, 3, 1, 1, residual=nhwc(r, dt), mask_src=nhwc(m, dt))
    assert rel_err(from_nhwc(dx), dx_ref) < TOL[dt]
    # wgrad un-fold: dW = dW_k * scale
    dwk = ops.conv_wgrad(nhwc(gy, dt), nhwc(x, dt), 3, 3, 1, 1)
    dW = ops.wgrad_finalize(dwk, sc, (Co, Ci, 3, 3), Ci)
    dW_ref = torch.nn.grad.conv2d_weight(x, (Co, Ci, 3, 3), gy, padding=1) * scale[:, None, None, None]
    assert rel_err(dW, dW_ref) < TOL[dt]


@pytest.mark.parametrize("cfg", [(3, 128, 16, 12, 128), (2, 256, 8, 10, 256), (2, 64, 6, 6, 192), (1, 512, 4, 6, 512)])
def test_stride2_3x3_dgrad_by_output_parity(cfg):
    """Input gradient of a 3x3 / stride 2 / pad 1 convolution on an even-sized input (the second conv of layer2.0 / 3.0 / 4.0): bf16 runs
    it as four parity-class forward-geometry launches over g (1x1 / 1x2 / 2x1 / 2x2 sub-filters taken as column blocks of w_dgrad,
    output stride 2) - against torch's conv2d_input, with the residual + mask epilogue, incl. the last row / column whose second tap
    falls outside g."""
    from tubedetr_amd import ops

    N, Co, Hg, Wg, Ci = cfg  # g [N, Co, Hg, Wg] -> dx [N, Ci, 2Hg, 2Wg]
    dt = torch.bfloat16
    H, W = 2 * Hg, 2 * Wg
    g = torch.Generator().manual_seed(21 + Co)
    w = rnd((Co, Ci, 3, 3), g, dt, 1.0 / math.sqrt(Ci * 9))
    _, wd, _, _ = ops.weight_prep(w.to(dev()), dt)
    gy, r, m = rnd((N, Co, Hg, Wg), g, dt), rnd((N, Ci, H, W), g, dt), rnd((N, Ci, H, W), g, dt)
    ref = (torch.nn.grad.conv2d_input((N, Ci, H, W), w.to(dev()), gy.to(dev()), stride=2, padding=1) + r.to(dev())) * (m.to(dev()) > 0)
    dx = ops.conv_dgrad(nhwc(gy, dt), wd, (H, W), 3, 3, 2, 1, residual=nhwc(r, dt), mask_src=nhwc(m, dt))
    assert rel_err(from_nhwc(dx), ref) < TOL[dt]
    dx2 = ops.conv_dgrad(nhwc(gy, dt), wd, (H, W), 3, 3, 2, 1)  # no epilogue operands
    assert rel_err(from_nhwc(dx2), torch.nn.grad.conv2d_input((N, Ci, H, W), w.to(dev()), gy.to(dev()), stride=2, padding=1)) < TOL[dt]


@pytest.mark.parametrize("dt", DT)
def test_strided_1x1_dgrad_scatter(dt):
    from tubedetr_amd import ops

    g = torch.Generator().manual_seed(3)
    N, Ci, H, W, Co = 2, 64, 9, 11, 128
    w = rnd((Co, Ci, 1, 1), g, dt, 0.1)
    gy = rnd((N, Co, 5, 6), g, dt)
    base = rnd((N, Ci, H, W), g, dt)
    m = rnd((N, Ci, H, W), g, dt)
    ref = (base * (m > 0) + torch.nn.grad.conv2d_input((N, Ci, H, W), w, gy, stride=2)) * (m > 0)
    _, wd, _, _ = ops.weight_prep(w.to(dev()), dt)
    dx = nhwc(base * (m > 0), dt)
    ops.conv1x1s_dgrad_scatter(nhwc(gy, dt), wd, dx, 2, mask_src=nhwc(m, dt))
    assert rel_err(from_nhwc(dx), ref) < TOL[dt]


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("shape", [(3775, 256, 768), (100, 256, 2048), (600, 2048, 256), (37, 768, 256), (600, 256, 8), (15100, 256, 3072)])
def test_linear_fwd_bwd(shape, dt):
    from tubedetr_amd import ops

    M, K, Nn = shape
    g = torch.Generator().manual_seed(4)
    x = rnd((M, K), g, dt)
    w = rnd((Nn, K), g, dt, 1 / math.sqrt(K))
    b = torch.randn(Nn, generator=g)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y_ref = F.relu(xr @ wr.t() + b)
    gy = rnd((M, Nn), g, dt)
    y_ref.backward(gy)
    wf, wd, _, _ = ops.weight_prep(w.to(dev()), dt)
    xd = x.to(dev(), dt)
    y = ops.linear_fwd(xd, wf, b.to(dev()), relu=True)
    assert rel_err(y, y_ref) < TOL[dt]
    gpre = ops.relu_bwd(gy.to(dev(), dt), y)
    dx = ops.linear_fwd(gpre, wd)
    assert rel_err(dx, xr.grad) < TOL[dt]
    dW = ops.linear_wgrad(gpre, xd)
    assert rel_err(dW, wr.grad) < TOL[dt]
    db = ops.colsum(gpre)
    assert rel_err(db, (gy * (y_ref > 0)).sum(0)) < TOL[dt]
    # bias gradient fused into the weight-gradient launch (accumulates: starts from a non-zero buffer here)
    db2 = torch.full((Nn,), 0.5, device=dev())
    dW2 = ops.linear_wgrad(gpre, xd, dbias=db2)
    assert rel_err(dW2, wr.grad) < TOL[dt]
    assert rel_err(db2 - 0.5, (gy * (y_ref > 0)).sum(0)) < TOL[dt]


@pytest.mark.parametrize("cfg", [(20001, 256, 1024, True, False, True), (70000, 64, 256, True, False, True), (40010, 128, 512, True, True, False),
                                 (33000, 192, 512, False, True, False), (140000, 256, 128, False, False, True), (30200, 256, 2048, False, False, True)])
def test_pointwise_persistent_instance_matches_tiled_math(cfg):
    """Shapes large enough to take the persistent weight-stationary instance (bf16, K <= 256, many rows): bias, residual,
    ReLU and ReLU-mask epilogues, ragged last M tile, against an fp32 matmul of the same bf16 inputs."""
    from tubedetr_amd import ops

    M, K, Nn, use_res, use_mask, relu = cfg
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(31)
    x = (torch.randn(M, K, generator=g)).to(dev(), dt)
    w = (torch.randn(Nn, K, generator=g) / math.sqrt(K)).to(dev(), dt)
    b = torch.randn(Nn, generator=g).to(dev())
    res = torch.randn(M, Nn, generator=g).to(dev(), dt) if use_res else None
    msk = torch.randn(M, Nn, generator=g).to(dev(), dt) if use_mask else None
    ref = x.float() @ w.float().t() + b
    if use_res:
        ref = ref + res.float()
    if relu:
        ref = F.relu(ref)
    if use_mask:
        ref = ref * (msk.float() > 0)
    y = ops.linear_fwd(x, w, b, residual=res, relu=relu, mask_src=msk)
    assert rel_err(y, ref) < TOL[dt]
    # dropout in the epilogue (the encoder's FFN in train mode): the mask of td_dropout over the same element indices
    yd = ops.linear_fwd(x, w, b, residual=res, relu=relu, mask_src=msk, dropout_p=0.1, seed=4321)
    ref_d = ops.dropout(ref.contiguous(), 0.1, 4321)
    sure = ref.abs() > 1e-3  # (an exact 0.0 of the fp32 reference can be a 1e-7 of another summation order)
    assert torch.equal((yd == 0)[sure], (ref_d == 0)[sure]) and 0.05 < ((yd == 0) & sure).float().mean().item() / max(sure.float().mean().item(), 1e-6) < 0.15
    assert rel_err(yd, ref_d) < TOL[dt]
    # in place on the residual (the dgrad chain accumulates into dx this way)
    if use_res:
        y2 = ops.linear_fwd(x, w, b, residual=res, relu=relu, mask_src=msk, out=res)
        assert torch.equal(y2, y)


@pytest.mark.parametrize("cfg", [(21, 90, 90, 256, 512, 2), (5, 181, 179, 128, 256, 2), (40, 45, 47, 512, 1024, 2), (12, 61, 64, 1024, 256, 2), (9, 100, 100, 64, 128, 3)])
def test_strided_pointwise_conv_on_every_route(cfg):
    """A 1x1 convolution with stride > 1 (the downsample branch of a stage's first block) in bf16: short K and many rows take the
    persistent instance (source pixel derived per row), K >= 512 the 256-row tiles through the one-tap form of the tap-uniform
    addressing, the rest the tiled kernel; odd extents (the last row / column is never read), ragged last tiles."""
    from tubedetr_amd import ops

    N, H, W, C, Co, st = cfg
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(77)
    x = torch.randn(N, H, W, C, generator=g).to(dev(), dt)
    w = (torch.randn(Co, C, 1, 1, generator=g) / math.sqrt(C)).to(dt).float().to(dev())
    b = torch.randn(Co, generator=g).to(dev())
    wf, _, b_out, _ = ops.weight_prep(w, dt, bias=b)
    y = ops.conv_fwd(x, wf, b_out, 1, 1, st, 0)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w, b, stride=st).permute(0, 2, 3, 1)
    assert y.shape == ref.shape
    assert rel_err(y, ref) < TOL[dt]


@pytest.mark.parametrize("cfg", [(41000, 512, 256, True, False, True), (41003, 1024, 128, False, True, False), (61000, 2048, 512, True, True, False),
                                 (41000, 576, 256, False, False, True)])
def test_pointwise_256_row_tiles(cfg):
    """bf16 pointwise layers with K >= 512 and enough rows take the eight-wavefront 256-row tile instance
    (conv_gemm_big_kernel, both tile widths): all epilogues, ragged last M tile, in-place residual."""
    from tubedetr_amd import ops

    M, K, Nn, use_res, use_mask, relu = cfg
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(41)
    x = (torch.randn(M, K, generator=g)).to(dev(), dt)
    w = (torch.randn(Nn, K, generator=g) / math.sqrt(K)).to(dev(), dt)
    b = torch.randn(Nn, generator=g).to(dev())
    res = torch.randn(M, Nn, generator=g).to(dev(), dt) if use_res else None
    msk = torch.randn(M, Nn, generator=g).to(dev(), dt) if use_mask else None
    ref = x.float() @ w.float().t() + b
    if use_res:
        ref = ref + res.float()
    if relu:
        ref = F.relu(ref)
    if use_mask:
        ref = ref * (msk.float() > 0)
    y = ops.linear_fwd(x, w, b, residual=res, relu=relu, mask_src=msk)
    assert rel_err(y, ref) < TOL[dt]
    if use_res:
        y2 = ops.linear_fwd(x, w, b, residual=res, relu=relu, mask_src=msk, out=res)
        assert torch.equal(y2, y)


@pytest.mark.parametrize("cfg", [(21, 64, 45, 44, 256, 1), (11, 128, 62, 61, 128, 1), (12, 128, 118, 117, 128, 2), (6, 256, 84, 83, 512, 1)])
def test_conv3x3_256_row_tiles(cfg):
    """3x3 layers large enough for the 256-row tile instance: forward (bias + ReLU, stride 1 and 2) and input gradient
    (residual + ReLU-mask), image borders and a ragged last tile, against torch's fp32 convolution of the same bf16 values."""
    from tubedetr_amd import ops

    N, Ci, H, W, Co, st = cfg
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(43)
    x = rnd((N, Ci, H, W), g, dt).to(dev())
    w = rnd((Co, Ci, 3, 3), g, dt, 1.0 / math.sqrt(9 * Ci)).to(dev())
    bias = torch.randn(Co, generator=g).to(dev())
    ref = F.relu(F.conv2d(x, w, bias, stride=st, padding=1))
    wf, wd, b_out, _ = ops.weight_prep(w, dt, bias=bias)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dt)
    y = ops.conv_fwd(xd, wf, b_out, 3, 3, st, 1, relu=True)
    assert rel_err(y.float().permute(0, 3, 1, 2), ref) < TOL[dt]
    if st == 1:
        gy = rnd(tuple(ref.shape), g, dt).to(dev())
        r = rnd((N, Ci, H, W), g, dt).to(dev())
        m = rnd((N, Ci, H, W), g, dt).to(dev())
        dx_ref = (torch.nn.grad.conv2d_input((N, Ci, H, W), w, gy, padding=1) + r) * (m > 0)
        to_rows = lambda t_: t_.permute(0, 2, 3, 1).contiguous().to(dt)
        dx = ops.conv_dgrad(to_rows(gy), wd, (H, W), 3, 3, 1, 1, residual=to_rows(r), mask_src=to_rows(m))
        assert rel_err(dx.float().permute(0, 3, 1, 2), dx_ref) < TOL[dt]


@pytest.mark.parametrize("dt", DT)
def test_linear_sigmoid_alpha_dropout(dt):
    from tubedetr_amd import ops

    g = torch.Generator().manual_seed(5)
    x, w = rnd((300, 256), g, dt), rnd((64, 256), g, dt, 0.06)
    y = ops.linear_fwd(x.to(dev(), dt), w.to(dev(), dt), None, sigmoid=True, alpha=0.5)
    assert rel_err(y, torch.sigmoid(0.5 * x @ w.t())) < TOL[dt]
    # dropout: kept elements are scaled by 1/(1-p), keep-rate ~ 1-p, mask is a pure function of (seed, index)
    y0 = ops.linear_fwd(x.to(dev(), dt), w.to(dev(), dt))
    y1 = ops.linear_fwd(x.to(dev(), dt), w.to(dev(), dt), dropout_p=0.25, seed=123)
    y2 = ops.linear_fwd(x.to(dev(), dt), w.to(dev(), dt), dropout_p=0.25, seed=123)
    assert torch.equal(y1, y2)
    kept = y1 != 0
    assert abs(kept.float().mean().item() - 0.75) < 0.02
    assert rel_err(y1[kept], (y0.float() / 0.75)[kept]) < 1e-2


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("shape", [(777, 256), (61, 768), (20011, 256), (33, 96)])  # vectorised bf16 instances (256 / 768 columns) and the generic one
def test_add_layernorm(shape, dt):
    from tubedetr_amd import ops

    g = torch.Generator().manual_seed(6)
    rows, cols = shape
    x, r = rnd((rows, cols), g, dt), rnd((rows, cols), g, dt)
    gamma, beta = torch.rand(cols, generator=g) + 0.5, torch.randn(cols, generator=g)
    sr = (x + r).requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y_ref = F.layer_norm(sr, (cols,), gr, br, 1e-5)
    dy = rnd((rows, cols), g, dt)
    extra = rnd((rows, cols), g, dt)
    y_ref.backward(dy)
    y, s, mean, rstd = ops.add_layernorm_fwd(x.to(dev(), dt), r.to(dev(), dt), gamma.to(dev()), beta.to(dev()), 1e-5)
    assert rel_err(y, y_ref) < TOL[dt]
    ds, dg, db = ops.add_layernorm_bwd(dy.to(dev(), dt), s, mean, rstd, gamma.to(dev()), extra.to(dev(), dt))
    s_used = s.float().cpu().requires_grad_(True)
    F.layer_norm(s_used, (cols,), gamma, beta, 1e-5).backward(dy)
    assert rel_err(ds, s_used.grad + extra) < TOL[dt]
    assert rel_err(dg, gr.grad) < TOL[dt] * 3 and rel_err(db, br.grad) < 1e-4


@pytest.mark.parametrize("dt", DT)
def test_maxpool_and_layout(dt):
    from tubedetr_amd import ops

    g = torch.Generator().manual_seed(7)
    x = torch.randn(3, 3, 21, 18, generator=g)
    xd = ops.nchw_to_nhwc(x.to(dev()), dt, 8)
    assert xd.shape == (3, 21, 18, 8)
    assert torch.equal(xd[..., :3].float().cpu(), x.to(dt).float().permute(0, 2, 3, 1)) and xd[..., 3:].abs().sum() == 0
    f = rnd((2, 64, 17, 20), g, dt)
    y = ops.maxpool3x3s2(nhwc(f, dt))
    assert torch.equal(from_nhwc(y), F.max_pool2d(f, 3, 2, 1))
    assert torch.equal(ops.nhwc_to_nchw(nhwc(f, dt)).cpu(), f)
    assert torch.equal(ops.cast(f.to(dev()), dt).float().cpu(), f)


@pytest.mark.parametrize("dt", DT)
def test_pos_sine_matches_oracle(dt):
    from oracle.tubedetr_oracle import pos_sine
    from tubedetr_amd import ops

    g = torch.Generator().manual_seed(8)
    mask = torch.rand(5, 11, 9, generator=g) > 0.7
    mask[:, 0, 0] = False
    mask[1] = False
    mask[2, :, 6:] = True
    ref = pos_sine(mask, 128).flatten(2).permute(0, 2, 1)  # (N, hw, 256)
    pos = ops.pos_sine(mask.to(dev()), 128, dt)
    tol = 2e-5 if dt == torch.float32 else 5e-3
    assert (pos.float().cpu() - ref).abs().max().item() < tol


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("cfg", [(3, 8, 151, 151, True), (1, 8, 100, 100, True), (7, 8, 1, 151, True), (2, 8, 37, 70, False),
                                 (2, 8, 20, 50, True), (2, 4, 70, 200, True), (40, 8, 151, 151, False), (1, 8, 33, 300, True)])
def test_mha_core_fwd_bwd(cfg, dt):
    from tubedetr_amd import ops

    B, H, Lq, Lk, use_mask = cfg
    E, hd = H * 32, 32
    g = torch.Generator().manual_seed(9)
    q, k, v = rnd((B, Lq, E), g, dt), rnd((B, Lk, E), g, dt), rnd((B, Lk, E), g, dt)
    kpm = (torch.rand(B, Lk, generator=g) > 0.8) if use_mask else None
    if kpm is not None:
        kpm[:, 0] = False
    scale = 1 / math.sqrt(hd)
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    qh = qr.view(B, Lq, H, hd).transpose(1, 2)
    kh = kr.view(B, Lk, H, hd).transpose(1, 2)
    vh = vr.view(B, Lk, H, hd).transpose(1, 2)
    sc = (qh @ kh.transpose(-1, -2)) * scale
    if kpm is not None:
        sc = sc.masked_fill(kpm[:, None, None, :], float("-inf"))
    pr = sc.softmax(-1)
    out_ref = (pr @ vh).transpose(1, 2).reshape(B, Lq, E)
    wavg_ref = pr.mean(1)
    dout = rnd((B, Lq, E), g, dt)
    dwavg = torch.randn(B, Lq, Lk, generator=g)
    (out_ref * dout).sum().add((wavg_ref * dwavg).sum()).backward()

    qd, kd, vd = (t.to(dev(), dt) for t in (q, k, v))
    out, probs, wavg = ops.mha_fwd(qd, kd, vd, kpm.to(dev()) if kpm is not None else None, H, scale, need_wavg=True)
    assert rel_err(out, out_ref) < TOL[dt]
    assert rel_err(wavg, wavg_ref) < 1e-4
    assert torch.equal(wavg.argmax(-1).cpu(), wavg_ref.argmax(-1))
    dq, dk, dv = ops.mha_bwd(qd, kd, vd, dout.to(dev(), dt), probs, dwavg.to(dev()), H, scale, torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd))
    for got, ref in ((dq, qr.grad), (dk, kr.grad), (dv, vr.grad)):
        assert rel_err(got, ref) < TOL[dt]


@pytest.mark.parametrize("cfg", [(3, 8, 151, 151, True), (2, 8, 100, 100, False), (7, 8, 1, 151, True), (2, 8, 37, 70, True), (2, 8, 60, 250, True)])
def test_mha_lean_matches_reference_and_probs_path(cfg):
    """The no-weights ("lean") attention core: nothing of size Lq x Lk is stored, the backward recomputes the probabilities
    from the row statistics.  Against an fp32 torch reference, and - with probability dropout on - against the probs-based
    kernels of the same library on the same seed (same mask by construction)."""
    from tubedetr_amd import ops

    dt = torch.bfloat16
    B, H, Lq, Lk, use_mask = cfg
    E, hd = H * 32, 32
    g = torch.Generator().manual_seed(19)
    q, k, v = rnd((B, Lq, E), g, dt), rnd((B, Lk, E), g, dt), rnd((B, Lk, E), g, dt)
    kpm = (torch.rand(B, Lk, generator=g) > 0.8) if use_mask else None
    if kpm is not None:
        kpm[:, 0] = False
    scale = 1 / math.sqrt(hd)
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    qh, kh, vh = (t.view(B, -1, H, hd).transpose(1, 2) for t in (qr, kr, vr))
    sc = (qh @ kh.transpose(-1, -2)) * scale
    if kpm is not None:
        sc = sc.masked_fill(kpm[:, None, None, :], float("-inf"))
    out_ref = (sc.softmax(-1) @ vh).transpose(1, 2).reshape(B, Lq, E)
    dout = rnd((B, Lq, E), g, dt)
    (out_ref * dout).sum().backward()
    qd, kd, vd = (t.to(dev(), dt) for t in (q, k, v))
    kd_ = kpm.to(dev()) if kpm is not None else None
    assert ops.mha_lean_ok(qd, kd, vd, H)
    out, stats, kp = ops.mha_lean_fwd(qd, kd, vd, kd_, H, scale)
    assert stats.shape == (B * H * Lq, 4)
    assert rel_err(out, out_ref) < TOL[dt]
    dq, dk, dv = ops.mha_lean_bwd(qd, kd, vd, kp, out, dout.to(dev(), dt), stats, H, scale, torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd))
    for got, ref in ((dq, qr.grad), (dk, kr.grad), (dv, vr.grad)):
        assert rel_err(got, ref) < TOL[dt]
    # dropout on: lean == probs-based path of the same seed
    p_drop, seed = 0.1, 1234
    out_a, stats_a, kp = ops.mha_lean_fwd(qd, kd, vd, kd_, H, scale, dropout_p=p_drop, seed=seed)
    out_b, probs_b, _ = ops.mha_fwd(qd, kd, vd, kd_, H, scale, dropout_p=p_drop, seed=seed)
    assert torch.equal(out_a, out_b)
    do = dout.to(dev(), dt)
    ga = ops.mha_lean_bwd(qd, kd, vd, kp, out_a, do, stats_a, H, scale, torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd), dropout_p=p_drop, seed=seed)
    gb = ops.mha_bwd(qd, kd, vd, do, probs_b, None, H, scale, torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd), dropout_p=p_drop, seed=seed)
    for a, b in zip(ga, gb):
        assert rel_err(a, b) < TOL[dt]


@pytest.mark.parametrize("cfg", [(3, 8, 151, 151), (1, 8, 100, 100), (5, 8, 1, 151), (2, 8, 40, 200)])
def test_mha_dropout_same_mask_in_both_dtypes(cfg):
    """The counter-based dropout mask is a function of (seed, element index) only: the exact-fp32 VALU kernels and the
    bf16 MFMA kernels must draw the same mask in forward and in both backward kernels."""
    from tubedetr_amd import ops

    B, H, Lq, Lk = cfg
    E = H * 32
    g = torch.Generator().manual_seed(19)
    q, k, v, dout = (rnd(sh, g, torch.bfloat16) for sh in ((B, Lq, E), (B, Lk, E), (B, Lk, E), (B, Lq, E)))
    dwavg = torch.randn(B, Lq, Lk, generator=g).to(dev())
    res = {}
    for dt in DT:
        qd, kd, vd, dd = (t.to(dev(), dt) for t in (q, k, v, dout))
        out, probs, wavg = ops.mha_fwd(qd, kd, vd, None, H, 0.2, need_wavg=True, dropout_p=0.3, seed=1234)
        dq, dk, dv = ops.mha_bwd(qd, kd, vd, dd, probs, dwavg, H, 0.2, torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd),
                                 dropout_p=0.3, seed=1234)
        res[dt] = (out, wavg, dq, dk, dv)
    for got, ref in zip(res[torch.bfloat16], res[torch.float32]):
        assert rel_err(got, ref) < 2e-2
    assert (res[torch.float32][1] == 0).float().mean().item() < 0.05  # wavg averages 8 independently dropped heads
    out_nodrop, _, _ = ops.mha_fwd(q.to(dev()), k.to(dev()), v.to(dev()), None, H, 0.2)
    assert rel_err(res[torch.bfloat16][0], out_nodrop) > 0.05  # dropout really applied on the MFMA path


def test_mha_strided_qkv_views():
    """q/k packed in one [B,L,2E] buffer (the fused QK projection output) and v separate."""
    from tubedetr_amd import ops

    g = torch.Generator().manual_seed(10)
    B, L, H, E = 2, 40, 8, 256
    qk = torch.randn(B, L, 2 * E, generator=g).to(dev())
    v = torch.randn(B, L, E, generator=g).to(dev())
    o1, p1, _ = ops.mha_fwd(qk[..., :E], qk[..., E:], v, None, H, 0.17)
    o2, p2, _ = ops.mha_fwd(qk[..., :E].contiguous(), qk[..., E:].contiguous(), v, None, H, 0.17)
    assert torch.equal(o1, o2) and torch.equal(p1, p2)
    qkb, vb = qk.bfloat16(), v.bfloat16()  # MFMA path: packed views keep 16-byte alignment
    o1, p1, _ = ops.mha_fwd(qkb[..., :E], qkb[..., E:], vb, None, H, 0.17)
    o2, p2, _ = ops.mha_fwd(qkb[..., :E].contiguous(), qkb[..., E:].contiguous(), vb, None, H, 0.17)
    assert torch.equal(o1, o2) and torch.equal(p1, p2)
    dq = torch.empty_like(qkb)
    dout = torch.randn(B, L, E, generator=g).to(dev()).bfloat16()
    ops.mha_bwd(qkb[..., :E], qkb[..., E:], vb, dout, p1, None, H, 0.17, dq[..., :E], dq[..., E:], torch.empty_like(vb))
    d2q, d2k, _ = ops.mha_bwd(qkb[..., :E].contiguous(), qkb[..., E:].contiguous(), vb, dout, p2, None, H, 0.17,
                              torch.empty_like(vb), torch.empty_like(vb), torch.empty_like(vb))
    assert torch.equal(dq[..., :E], d2q) and torch.equal(dq[..., E:], d2k)


def test_dropout_masks_of_consecutive_steps_are_independent():
    """The device step counter re-keys the counter-based RNG through a hash: the mask of step c must not be the step-0
    mask shifted by c elements (what a linear seed + c * golden re-keying produces), and masks of different steps must
    be uncorrelated."""
    from tubedetr_amd import ops

    n, p = 1 << 16, 0.5
    x = torch.ones(n, device=dev())
    ctr = torch.zeros(1, dtype=torch.int32, device=dev())
    masks = []
    try:
        ops.set_dropout_counter(ctr)
        for c in range(4):
            ctr.fill_(c)
            masks.append(ops.dropout(x, p, 1234) > 0)
        again = ops.dropout(x, p, 1234) > 0
    finally:
        ops.set_dropout_counter(None)
    assert torch.equal(masks[3], again)  # same (seed, counter) -> same mask (backward regenerates it)
    for c in range(1, 4):
        m0, mc = masks[0], masks[c]
        assert abs(mc.float().mean().item() - (1 - p)) < 0.02
        for shift in (0, c, -c):  # a shifted copy would agree on ~100 % of the overlap; independent masks on ~50 %
            a = m0[max(0, shift) : n + min(0, shift)]
            b = mc[max(0, -shift) : n - max(0, shift)]
            agree = (a == b).float().mean().item()
            assert 0.45 < agree < 0.55, (c, shift, agree)


@pytest.mark.parametrize("dt", DT)
def test_pos_sine_writes_zero_text_rows(dt):
    """rows_per_image > h*w: the rows behind the visual tokens are zeros (the text tokens' positional operand, transformer.py:323-326)."""
    from oracle.tubedetr_oracle import pos_sine as pos_ref
    from tubedetr_amd import ops

    g = torch.Generator().manual_seed(4)
    mask = torch.rand(5, 7, 9, generator=g) > 0.7
    mask[:, 0, 0] = False
    S = 7 * 9 + 11
    got = ops.pos_sine(mask.to(dev()), 128, dt, rows=S).float().cpu()
    ref = pos_ref(mask, 128).flatten(2).permute(0, 2, 1)  # (N, hw, C)
    assert got.shape == (5, S, 256)
    assert (got[:, : 7 * 9] - ref).abs().max().item() < (1e-5 if dt == torch.float32 else 1e-2)
    assert got[:, 7 * 9 :].abs().max().item() == 0


LINEAR_EX = [
    # M (mapped rows), rows1, rows2, K, N, maps?, residual?, shared weight?
    (300, 77, 300, 256, 512, True, False, True),
    (1000, 1000, 1000, 256, 256, False, False, True),
    (40000, 9000, 40000, 256, 256, True, True, True),
    (513, 200, 513, 64, 192, True, True, False),
    (2000, 2000, 0, 256, 48, False, True, False),
]


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("cfg", LINEAR_EX)
def test_linear_ex_two_sources_and_row_maps(cfg, dt):
    """td_linear_ex: [A1[a1_map] | A2] @ W^T (+ bias + residual[res_map]) written to out[out_map], shared and concatenated weights,
    against the same arithmetic in fp32 torch on materialised operands."""
    from tubedetr_amd import ops

    M, rows1, rows2, K, N, mapped, with_res, shared = cfg
    g = torch.Generator().manual_seed(M + N)
    a1 = rnd((rows1, K), g, dt)
    a2 = rnd((rows2, K), g, dt) if rows2 else None
    w = rnd((N, K if (shared or a2 is None) else 2 * K), g, dt, 1.0 / math.sqrt(K))
    bias = torch.randn(N, generator=g)
    a1_map = torch.randint(0, rows1, (M,), generator=g, dtype=torch.int32) if mapped else None
    out_rows = M + 17 if mapped else M
    out_map = torch.randperm(out_rows, generator=g)[:M].to(torch.int32) if mapped else None
    res = rnd((rows1 if mapped else M, N), g, dt) if with_res else None
    res_map = a1_map if (mapped and with_res) else None
    X1 = a1[a1_map.long()] if mapped else a1[:M]
    ref = X1 @ (w[:, :K]).t() + bias
    if a2 is not None:
        ref = ref + a2[:M] @ (w[:, :K] if shared else w[:, K:]).t()
    if with_res:
        ref = ref + (res[res_map.long()] if res_map is not None else res[:M])
    ref = ref.relu()
    d = dev()
    to = lambda t_: None if t_ is None else t_.to(d, dt if t_.is_floating_point() else t_.dtype)
    out = torch.full((out_rows, N), 7.0, dtype=dt, device=d)
    ops.linear_ex(to(a1), to(w), bias.to(d), a2=to(a2), a1_map=to(a1_map), w_shared=shared, out=out, out_map=to(out_map), residual=to(res), res_map=to(res_map), relu=True)
    got = out.float().cpu()
    got_rows = got[out_map.long()] if mapped else got
    assert rel_err(got_rows, ref) < TOL[dt], cfg
    if mapped:  # rows no index points at stay untouched
        untouched = torch.ones(out_rows, dtype=torch.bool)
        untouched[out_map.long()] = False
        assert (got[untouched] == 7.0).all()


@pytest.mark.parametrize("dt", DT)
def test_rows_copy_and_segment_sum(dt):
    from tubedetr_amd import ops

    g = torch.Generator().manual_seed(12)
    d = dev()
    src = rnd((50, 256), g, dt)
    add = rnd((120, 256), g, dt)
    smap = torch.randint(0, 50, (120,), generator=g, dtype=torch.int32)
    dmap = torch.randperm(200, generator=g)[:120].to(torch.int32)
    dst = torch.full((200, 256), 3.0, dtype=dt, device=d)
    ops.rows_copy(src.to(d, dt), smap.to(d), dst, dmap.to(d), 120)
    got = dst.float().cpu()
    assert torch.equal(got[dmap.long()], src[smap.long()])
    dst2 = torch.empty((120, 256), dtype=dt, device=d)
    ops.rows_copy(src.to(d, dt), smap.to(d), dst2, None, 120, add=add.to(d, dt))
    assert rel_err(dst2.float().cpu(), src[smap.long()] + add) < TOL[dt]
    # segment sums with ragged segments (incl. an empty one) and a scattered output
    lens = torch.tensor([3, 0, 5, 1, 7, 2])
    ptr = torch.zeros(7, dtype=torch.int32)
    ptr[1:] = torch.cumsum(lens, 0).to(torch.int32)
    idx = torch.randint(0, 120, (int(lens.sum()),), generator=g, dtype=torch.int32)
    omap = torch.tensor([4, 0, 9, 2, 7, 5], dtype=torch.int32)
    out = torch.full((10, 256), 5.0, dtype=dt, device=d)
    ops.rows_segment_sum(add.to(d, dt), idx.to(d), ptr.to(d), out, omap.to(d))
    got = out.float().cpu()
    for r in range(6):
        want = add[idx[ptr[r] : ptr[r + 1]].long()].sum(0)
        assert (got[omap[r]] - want).abs().max().item() <= TOL[dt] * max(1.0, want.abs().max().item()), r
    assert (got[[1, 3, 6, 8]] == 5.0).all()


@pytest.mark.parametrize("dt", DT)
def test_replication_and_slow_fast_aggregation_match_indexing(dt):
    """functional.SlowFastAggregateFn / ReplicateRowsFn (the temporal replication as an index inside its consumers) against the
    reference's formulation on materialised tensors (transformer.py:393-445): forward and every gradient."""
    from tubedetr_amd import functional as Fk

    g = torch.Generator().manual_seed(31)
    d_, hw, L, k = 256, 6, 3, 2
    durations = [5, 4]
    b, t = len(durations), max(durations)
    n_clips = math.ceil(t / k)
    n, S, F_ = b * n_clips, hw + L, b * t
    owner = (torch.arange(b)[:, None] * n_clips + torch.arange(t)[None, :] // k).reshape(-1)
    mem = rnd((n * S, d_), g, dt)
    fast = rnd((F_ * hw, d_), g, dt)
    W = rnd((d_, d_), g, dt, 1.0 / 16)
    bias = torch.randn(d_, generator=g)
    gy = rnd((F_ * S, d_), g, dt)
    # reference formulation (fp32, materialised)
    mr, fr, Wr, br = mem.clone().requires_grad_(), fast.clone().requires_grad_(), W.clone().requires_grad_(), bias.clone().requires_grad_()
    frames = mr.view(n, S, d_)[owner]
    vis = frames[:, :hw]
    out_ref = torch.cat([vis + F.linear(vis + fr.view(F_, hw, d_), Wr, br), frames[:, hw:]], 1).reshape(F_ * S, d_)
    out_ref.backward(gy)
    dvc = dev()
    maps = Fk.ReplicaMaps(owner, n, hw, L, dvc)
    m2, f2 = mem.to(dvc, dt).requires_grad_(), fast.to(dvc, dt).requires_grad_()
    W2, b2 = W.to(dvc).requires_grad_(), bias.to(dvc).requires_grad_()
    out = Fk.SlowFastAggregateFn.apply(m2, f2, W2, b2, maps)
    assert rel_err(out.float().cpu(), out_ref) < TOL[dt]
    out.backward(gy.to(dvc, dt))
    torch.cuda.synchronize()
    for name, got, want in (("mem", m2.grad, mr.grad), ("fast", f2.grad, fr.grad), ("W", W2.grad, Wr.grad), ("b", b2.grad, br.grad)):
        assert rel_err(got.float().cpu(), want) < 2 * TOL[dt], name
    # --no_fast: plain replication
    m3 = mem.to(dvc, dt).requires_grad_()
    rep = Fk.ReplicateRowsFn.apply(m3, maps)
    mr2 = mem.clone().requires_grad_()
    rep_ref = mr2.view(n, S, d_)[owner].reshape(F_ * S, d_)
    assert torch.equal(rep.float().cpu(), rep_ref.detach())
    rep.backward(gy.to(dvc, dt))
    rep_ref.backward(gy)
    assert rel_err(m3.grad.float().cpu(), mr2.grad) < TOL[dt]


def _torch_cross_attention(tgt, qpos, mem, pos, W_in, b_in, W_out, b_out, key_pad, F_, S, H):
    """nn.MultiheadAttention's arithmetic for one query per frame (models/transformer.py:725-745), plain fp32 torch."""
    E = tgt.shape[1]
    hd = E // H
    q = (tgt + qpos) @ W_in[:E].t() + b_in[:E]
    k = ((mem + pos) @ W_in[E : 2 * E].t() + b_in[E : 2 * E]).view(F_, S, H, hd)
    v = (mem @ W_in[2 * E :].t() + b_in[2 * E :]).view(F_, S, H, hd)
    sc = torch.einsum("fhd,fshd->fhs", q.view(F_, H, hd) / math.sqrt(hd), k)
    sc = sc.masked_fill(key_pad[:, None, :], float("-inf"))
    pr = sc.softmax(-1)
    ctxv = torch.einsum("fhs,fshd->fhd", pr, v).reshape(F_, E)
    return ctxv @ W_out.t() + b_out, pr.mean(1).view(F_, 1, S)


@pytest.mark.parametrize("cfg", [(40, 151), (7, 69), (3, 13), (130, 36)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_cross_attention_with_query_side_projections(cfg, dt):
    """functional.CrossQ1Fn (the decoder's time-aligned cross-attention without key / value projections of the memory, csrc/cross_attn.hip)
    against nn.MultiheadAttention's own formulation in fp32 torch: output, head-averaged weights, and the gradients of the query
    input, the memory, and every parameter - with a loss on the weights too (the guided-attention term), key padding, three layers
    sharing one memory (the d(memory) accumulation across layers)."""
    from tubedetr_amd import functional as Fk

    F_, S = cfg
    E, H, nl = 256, 8, 3
    g = torch.Generator().manual_seed(5 + S)
    r = lambda *sh, s=1.0: (torch.randn(*sh, generator=g) * s).to(dt).float().to(dev())
    tgt, qpos, mem, pos = r(F_, E), r(F_, E), r(F_ * S, E), r(F_ * S, E)
    key_pad = (torch.rand(F_, S, generator=g) < 0.2).to(dev())
    key_pad[:, 0] = False
    params = [[r(3 * E, E, s=1 / 16), r(3 * E, s=0.5), r(E, E, s=1 / 16), r(E, s=0.5)] for _ in range(nl)]
    wo, ww = r(nl, F_, E), r(nl, F_, 1, S)

    def run(fn, leafs_dtype):
        t_, m_ = tgt.clone().requires_grad_(True), mem.clone().requires_grad_(True)
        ps = [[p.clone().requires_grad_(True) for p in layer] for layer in params]
        loss = fn(t_.to(leafs_dtype), m_.to(leafs_dtype), ps)
        loss.backward()
        return loss.detach(), [t_.grad, m_.grad] + [p.grad for layer in ps for p in layer]

    mag = [0.0]  # sum of the magnitudes of the loss terms: what a relative bound on the (cancelling) sum refers to

    def ref(t_, m_, ps):
        loss = 0.0
        x = t_
        for l in range(nl):
            o, w = _torch_cross_attention(x, qpos, m_.view(F_ * S, E), pos, *ps[l], key_pad, F_, S, H)
            loss = loss + (o * wo[l]).sum() + (w * ww[l]).sum() * 30
            mag[0] += ((o * wo[l]).abs().sum() + (w * ww[l]).abs().sum() * 30).item()
            x = t_ + 0.1 * o  # the next layer's query depends on this layer's output
        return loss

    def new(t_, m_, ps):
        loss = 0.0
        anchor = Fk.cross_q1_memory(m_, pos.to(dt))
        x = t_
        for l in range(nl):
            o, w = Fk.multihead_attention_q1(x, anchor, *ps[l], key_pad, F_, S, H, need_weights=True, q_pos=qpos.to(dt))
            loss = loss + (o.float() * wo[l]).sum() + (w * ww[l]).sum() * 30
            x = t_ + (0.1 * o.float()).to(dt)
        return loss

    l_ref, g_ref = run(ref, torch.float32)
    l_new, g_new = run(new, dt)
    tol = 2e-4 if dt == torch.float32 else 3e-2
    assert abs(l_new - l_ref).item() <= (2e-5 if dt == torch.float32 else 2e-3) * mag[0]
    names = ["tgt", "mem"] + [f"layer{l}.{n}" for l in range(nl) for n in ("in_proj_weight", "in_proj_bias", "out_proj.weight", "out_proj.bias")]
    for n, a, b in zip(names, g_new, g_ref):
        if n.endswith("in_proj_bias"):  # the key bias shifts every score of a row equally: zero gradient (fp32 noise in the reference)
            assert a[E : 2 * E].abs().max().item() == 0.0 and b[E : 2 * E].abs().max().item() < 1e-3 * b.abs().max().item()
        assert rel_err(a, b) < tol, n


def test_cross_attention_query_side_with_a_memory_that_needs_no_gradient():
    """memory.requires_grad = False (a frozen encoder): td_cross_q1_bwd gets d_mem = NULL - no [F*S, E] fp32 buffer is formed per
    layer - and the query / parameter gradients equal those of the run that does differentiate the memory."""
    from tubedetr_amd import functional as Fk

    F_, S, E, H = 24, 151, 256, 8
    g = torch.Generator().manual_seed(9)
    r = lambda *sh, s=1.0: (torch.randn(*sh, generator=g) * s).to(dev())
    tgt, qpos, mem, pos = r(F_, E), r(F_, E), r(F_ * S, E), r(F_ * S, E)
    ps0 = [r(3 * E, E, s=1 / 16), r(3 * E, s=0.5), r(E, E, s=1 / 16), r(E, s=0.5)]
    wo = r(F_, E)
    out = []
    for mem_grad in (True, False):
        peaks = []
        for _rep in range(2):  # (the smaller peak of two runs: a run may be the one in which the zero-fill arena takes a new 128 MB chunk)
            t_ = tgt.clone().requires_grad_(True)
            m_ = mem.clone().requires_grad_(mem_grad)
            ps = [p.clone().requires_grad_(True) for p in ps0]
            anchor = Fk.cross_q1_memory(m_, pos)
            x = t_
            loss = 0.0
            for _ in range(2):
                o, _w = Fk.multihead_attention_q1(x, anchor, *ps, None, F_, S, H, need_weights=False, q_pos=qpos)
                loss = loss + (o * wo).sum()
                x = t_ + 0.1 * o
            before = torch.cuda.memory_allocated()
            torch.cuda.reset_peak_memory_stats()
            loss.backward()
            peaks.append(torch.cuda.max_memory_allocated() - before)
        out.append(([t_.grad] + [p.grad for p in ps], min(peaks), m_.grad))
    (g1, peak1, mg1), (g0, peak0, mg0) = out
    assert mg1 is not None and mg0 is None
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)
    assert peak1 - peak0 >= F_ * S * E * 4  # the fp32 d(memory) buffer exists only when the memory wants it


@pytest.mark.parametrize("F_,S,nl", [(24, 151, 6), (5, 37, 3), (3, 16, 1)])
def test_cross_attention_deferred_memory_gradient(F_, S, nl):
    """bf16 mode: td_cross_q1_bwd_coef + ONE td_cross_q1_dmem (coefficients per memory row, then a [S][16 nl] x [16 nl][E] MFMA product
    per frame) against td_cross_q1_bwd's fp32 accumulation over the layers (the VALU kernel: an fp32 d(memory) keeps it off the matrix
    pipe) - d_u and the memory gradient within bf16 rounding of the probabilities / coefficients and of the stored result; dropout and key
    padding on; a layer that never ran contributes nothing."""
    from tubedetr_amd import ops

    E, H = 256, 8
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(100 + S)
    r = lambda *sh, s=1.0: (torch.randn(*sh, generator=g) * s).to(dt).to(dev())
    mem, pos = r(F_ * S, E), r(F_ * S, E)
    key_pad = (torch.rand(F_, S, generator=g) < 0.2).to(dev())
    key_pad[:, 0] = False
    ran = [l for l in range(nl) if not (nl == 6 and l == 4)]  # (layer 4 of the six never runs its backward)
    KP = (16 * nl + 31) // 32 * 32
    coef = torch.zeros((F_ * S, KP), dtype=dt, device=dev())
    dmem_ref, layers, first = torch.empty((F_ * S, E), dtype=torch.float32, device=dev()), [None] * nl, True
    for l in ran:
        u = r(F_, H * E, s=0.2)
        seed = 1234 + l
        probs, _wavg, _zext = ops.cross_q1_fwd(u, mem, pos, key_pad, F_, S, H, need_wavg=True, dropout_p=0.1, seed=seed)
        d_zext = r(F_, H * E + H, s=0.5)
        dwa = (torch.randn(F_, S, generator=g) * 0.3).to(dev())
        du_ref = ops.cross_q1_bwd(u, mem, pos, probs, d_zext, dwa, dmem_ref, not first, F_, S, H, dropout_p=0.1, seed=seed)
        du_new = ops.cross_q1_bwd_coef(u, mem, pos, probs, d_zext, dwa, coef, 16 * l, F_, S, H, dropout_p=0.1, seed=seed)
        assert rel_err(du_new, du_ref) < 1e-2
        layers[l] = (u, d_zext)
        first = False
    dmem = ops.cross_q1_dmem(coef, layers, F_, S, H, E)
    assert dmem.dtype == dt and dmem.shape == (F_ * S, E) and torch.isfinite(dmem.float()).all()
    assert rel_err(dmem, dmem_ref) < 1e-2
    # per-row check too (a wrong row / channel mapping hides behind a max-norm over the tensor when one row dominates)
    num = (dmem.float() - dmem_ref).norm(dim=1)
    den = dmem_ref.norm(dim=1).clamp_min(1e-3 * dmem_ref.norm(dim=1).max())
    assert (num / den).max().item() < 2e-2


def test_cross_attention_query_side_draws_the_same_dropout_mask_as_the_projected_path():
    """Same (seed, element index) dropout keys as td_mha_fwd with Lq = 1: with dropout on, the query-side formulation and the
    projected-memory path (functional.MHAFn) agree on outputs, returned weights and gradients."""
    from tubedetr_amd import functional as Fk

    F_, S, E, H = 24, 151, 256, 8
    dt = torch.float32
    g = torch.Generator().manual_seed(11)
    r = lambda *sh, s=1.0: (torch.randn(*sh, generator=g) * s).to(dev())
    tgt, mem = r(F_, E), r(F_ * S, E)
    W_in, b_in, W_out, b_out = r(3 * E, E, s=1 / 16), r(3 * E, s=0.5), r(E, E, s=1 / 16), r(E, s=0.5)
    key_pad = (torch.rand(F_, S, generator=g) < 0.2).to(dev())
    key_pad[:, 0] = False
    wo, ww = r(F_, E), r(F_, 1, S)
    res = []
    for which in ("q1", "projected"):
        torch.manual_seed(77)  # both paths draw their dropout seeds from the generator keyed by torch's seed
        Fk._SEED_STATE["torch_seed"] = None
        t_, m_ = tgt.clone().requires_grad_(True), mem.clone().requires_grad_(True)
        ps = [p.clone().requires_grad_(True) for p in (W_in, b_in, W_out, b_out)]
        Fk.set_wgrad_deferral(which != "q1")  # ...and the immediate (not deferred) weight-gradient branch of the new node, as under torch DDP
        try:
            if which == "q1":
                o, w = Fk.multihead_attention_q1(t_, Fk.cross_q1_memory(m_, None), *ps, key_pad, F_, S, H, need_weights=True, attn_dropout=0.3, training=True)
            else:
                o, w = Fk.multihead_attention(t_, m_, m_, *ps, key_pad, F_, 1, S, H, True, attn_dropout=0.3, training=True)
            ((o * wo).sum() + (w * ww).sum() * 30).backward()
        finally:
            Fk.set_wgrad_deferral(True)
        res.append([o.detach(), w.detach(), t_.grad, m_.grad] + [p.grad for p in ps])
    assert (res[0][1] == 0).float().mean().item() < 0.35 and rel_err(res[0][1], res[1][1]) < 1e-4  # same dropped entries
    for a, b in zip(res[0], res[1]):
        assert rel_err(a, b) < 5e-4
