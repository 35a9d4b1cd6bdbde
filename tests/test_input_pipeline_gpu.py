"""Device-side input path (tubedetr_amd/data.py + td_frames_to_nhwc): uint8 clips sent once, normalisation / layout / the
slow-fast split done by the trunk's input kernel, against the reference's format (normalised fp32 frames, the slow clip
as a second tensor, util/misc.py:106-178 + datasets/vidstg.py:250-251): same outputs, same losses, same gradients."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_frames_to_nhwc_u8_index_matches_torch():
    from tubedetr_amd import _hip
    import ctypes as C

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    vid = torch.randint(0, 256, (7, 3, 10, 12), generator=g, dtype=torch.uint8).to(dev)
    idx = torch.tensor([0, 3, 6, 2], dtype=torch.int32, device=dev)
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    for dt, cpad in ((torch.float32, 4), (torch.bfloat16, 8)):
        out = torch.empty((4 + 7, 10, 12, cpad), dtype=dt, device=dev)
        srcs = (_hip.FrameSource * 2)()
        srcs[0].data, srcs[0].dtype, srcs[0].n, srcs[0].index = vid.data_ptr(), _hip.TD_U8, 4, idx.data_ptr()
        srcs[1].data, srcs[1].dtype, srcs[1].n, srcs[1].index = vid.data_ptr(), _hip.TD_U8, 7, None
        _hip.check(_hip.lib().td_frames_to_nhwc(srcs, 2, 3, 10, 12, cpad, (C.c_float * 3)(*mean), (C.c_float * 3)(*[1 / s for s in std]),
                                                out.data_ptr(), _hip.dtype_code(dt), _hip.stream_ptr()), "td_frames_to_nhwc")
        ref = (vid.float() / 255 - torch.tensor(mean, device=dev).view(1, 3, 1, 1)) / torch.tensor(std, device=dev).view(1, 3, 1, 1)
        ref = torch.cat([ref[idx.long()], ref]).permute(0, 2, 3, 1)
        tol = 1e-6 if dt == torch.float32 else 1e-2
        assert (out[..., :3].float() - ref).abs().max().item() < tol * 3
        assert out[..., 3:].abs().max().item() == 0


def test_uint8_pipeline_equals_reference_format_step():
    import tubedetr_amd
    from oracle.weights import fill_state, state_spec
    from oracle.tubedetr_oracle import OracleConfig
    from tubedetr_amd.data import ClipPipeline
    from tubedetr_amd.harness import FixedTokenizer, forward_step
    from tubedetr_amd.models import build_model

    dev = torch.device("cuda:0")
    k, T, res, L = 2, 6, 64, 5
    cfg = OracleConfig(stride=k)
    sd = fill_state(state_spec(cfg), 3)
    model, criterion, wd = build_model(tubedetr_amd.default_args(stride=k, compute_dtype=torch.float32))
    model.load_state_dict(sd, strict=True)
    model.to(dev).eval()
    g = torch.Generator().manual_seed(1)
    videos = [torch.randint(0, 256, (T, 3, res, res), generator=g, dtype=torch.uint8), torch.randint(0, 256, (T, 3, res, res), generator=g, dtype=torch.uint8)]
    ids = torch.randint(3, 50000, (2, L), generator=g)
    ids[:, 0], ids[:, -1] = 0, 2
    att = torch.ones(2, L, dtype=torch.long)
    boxes = torch.cat([torch.rand(2 * T, 2, generator=g) * 0.5 + 0.25, torch.rand(2 * T, 2, generator=g) * 0.3 + 0.1], 1)
    inter = [[0, T - 1], [0, T - 1]]
    model.transformer.tokenizer = FixedTokenizer(ids, att)
    # (a) the new path
    pipe = ClipPipeline(dev, k)
    batch = pipe.collect(pipe.stage(videos, ids, att, boxes, inter))
    params = [p for p in model.parameters() if p.requires_grad]
    loss_a, ld_a, out_a, _ = forward_step(model, criterion, wd, batch)
    loss_a.backward()
    ga = [None if p.grad is None else p.grad.clone() for p in params]
    for p in params:
        p.grad = None
    # (b) the reference's format: normalised fp32 frames, slow clip as a second tensor
    body = model.backbone[0].body
    mean = torch.tensor(body.pixel_mean).view(1, 3, 1, 1)
    std = torch.tensor(body.pixel_std).view(1, 3, 1, 1)
    fast = torch.cat([(v.float() / 255 - mean) / std for v in videos]).to(dev)
    slow = torch.cat([fast[i * T : (i + 1) * T][::k] for i in range(2)])
    ref = {"frames": slow, "frames_mask": torch.zeros(slow.shape[0], res, res, dtype=torch.bool, device=dev), "frames_fast": fast,
           "fast_mask": torch.zeros(fast.shape[0], res, res, dtype=torch.bool, device=dev), "durations": [T, T], "input_ids": ids, "attention_mask": att,
           "target_boxes": boxes.to(dev), "inter_idx": inter}
    loss_b, ld_b, out_b, _ = forward_step(model, criterion, wd, ref)
    loss_b.backward()
    assert abs(loss_a.item() - loss_b.item()) < 1e-4 * abs(loss_b.item())
    for key in ("pred_boxes", "pred_sted"):
        assert (out_a[key] - out_b[key]).abs().max().item() < 1e-4
    scale = max(p.grad.norm().item() for p in params if p.grad is not None)  # numerically-zero gradients (softmax-invariant biases) are judged on the model's scale
    for p, a in zip(params, ga):
        if a is None:
            assert p.grad is None
            continue
        # the two paths round the normalised pixels differently in the last bit ((x * (1/255) - m) * (1/s) vs (x / 255 - m) / s);
        # 104 convolutions and the min / max / sign kinks of the losses amplify that to ~1e-3 .. 1e-2 on individual gradient entries,
        # so the comparison is per parameter in norm (a wrong frame order or normalisation constant would be O(1))
        assert (a - p.grad).norm().item() <= 2e-2 * p.grad.norm().item() + 1e-5 * scale


def test_ragged_batch_padding_is_zero_after_normalisation():
    """Videos of different H x W in one batch: the reference normalises FIRST and pads with zeros
    (NestedTensor.from_tensor_list, util/misc.py:158-170), so padded pixels are exactly 0 in the trunk's input; the uint8
    pipeline pads raw pixels and lets td_frames_to_nhwc zero the padded area through the per-frame extents."""
    import ctypes as C

    from tubedetr_amd import _hip
    from tubedetr_amd.data import ClipPipeline
    from tubedetr_amd.util.misc import NestedTensor

    dev = torch.device("cuda:0")
    k = 2
    g = torch.Generator().manual_seed(4)
    videos = [torch.randint(1, 256, (4, 3, 20, 32), generator=g, dtype=torch.uint8), torch.randint(1, 256, (6, 3, 28, 24), generator=g, dtype=torch.uint8)]
    ids = torch.randint(3, 50000, (2, 5), generator=g)
    att = torch.ones(2, 5, dtype=torch.long)
    boxes = torch.rand(10, 4, generator=g)
    pipe = ClipPipeline(dev, k)
    batch = pipe.collect(pipe.stage(videos, ids, att, boxes, [[0, 3], [0, 5]]))
    torch.cuda.synchronize()
    mean = torch.tensor((0.485, 0.456, 0.406)).view(1, 3, 1, 1)
    std = torch.tensor((0.229, 0.224, 0.225)).view(1, 3, 1, 1)
    # the reference's collate: normalised clips (C, T, H, W), zero-padded
    ref = NestedTensor.from_tensor_list([((v.float() / 255 - mean) / std).transpose(0, 1) for v in videos])
    H, W = ref.tensors.shape[-2:]
    assert (H, W) == (28, 32)
    assert torch.equal(batch["fast_mask"].cpu(), ref.mask)
    slow_ref = torch.cat([ref.tensors[0:4:k], ref.tensors[4:10:k]])
    assert torch.equal(batch["frames_mask"].cpu(), torch.cat([ref.mask[0:4:k], ref.mask[4:10:k]]))
    for name, want in (("frames_fast", ref.tensors), ("frames", slow_ref)):
        fs = batch[name]
        n = fs.n_frames
        assert n == want.shape[0]
        for dt, cpad, tol in ((torch.float32, 4, 3e-6), (torch.bfloat16, 8, 3e-2)):
            out = torch.full((n, H, W, cpad), float("nan"), dtype=dt, device=dev)
            srcs = (_hip.FrameSource * len(fs.parts))()
            for s_, (t_, idx), vhw in zip(srcs, fs.parts, fs.valid):
                assert vhw is not None
                s_.data, s_.dtype, s_.n = t_.data_ptr(), _hip.TD_U8, (idx.numel() if idx is not None else t_.shape[0])
                s_.index, s_.valid_hw = (idx.data_ptr() if idx is not None else None), vhw.data_ptr()
            _hip.check(_hip.lib().td_frames_to_nhwc(srcs, len(fs.parts), 3, H, W, cpad, (C.c_float * 3)(0.485, 0.456, 0.406),
                                                    (C.c_float * 3)(*[1 / s for s in (0.229, 0.224, 0.225)]), out.data_ptr(), _hip.dtype_code(dt),
                                                    _hip.stream_ptr()), "td_frames_to_nhwc")
            got = out[..., :3].float().cpu().permute(0, 3, 1, 2)
            assert (got - want).abs().max().item() < tol
            pad = (want == 0).all(1)  # padded pixels: exactly zero, not (0 - mean) / std
            assert pad.any() and (got.permute(0, 2, 3, 1)[pad] == 0).all()
