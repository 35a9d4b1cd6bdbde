"""Boundary carrier type of the hot path: the reference's ``util.misc.NestedTensor`` (util/misc.py:106-178)
- a batch of frames plus its pad mask (True = padded pixel).  Written from the interface description in
SURVEY.md section 8(a) F1; only the members the hot path and its callers touch."""
from __future__ import annotations

from typing import List, Optional

import torch


class NestedTensor(object):
    def __init__(self, tensors: torch.Tensor, mask: Optional[torch.Tensor]):
        self.tensors = tensors
        self.mask = mask

    def to(self, *args, **kwargs) -> "NestedTensor":
        m = self.mask.to(*args, **kwargs) if self.mask is not None else None
        return type(self)(self.tensors.to(*args, **kwargs), m)

    def decompose(self):
        return self.tensors, self.mask

    @classmethod
    def from_tensor_list(cls, tensor_list: List[torch.Tensor], do_round: bool = False) -> "NestedTensor":
        """Images (C,H,W) -> (B,C,Hmax,Wmax); video clips (C,T,H,W) -> all frames (sum T, C, Hmax, Wmax).
        Pixels outside an item's own extent are zero and masked True."""
        first = tensor_list[0]
        if first.ndim not in (3, 4):
            raise ValueError("not supported")
        video = first.ndim == 4
        hs = [t.shape[-2] for t in tensor_list]
        ws = [t.shape[-1] for t in tensor_list]
        H, W = max(hs), max(ws)
        if do_round:
            H, W = -(-H // 128) * 128, -(-W // 128) * 128
        C = max(t.shape[0] for t in tensor_list)
        n = sum(t.shape[1] for t in tensor_list) if video else len(tensor_list)
        out = torch.zeros((n, C, H, W), dtype=first.dtype, device=first.device)
        mask = torch.ones((n, H, W), dtype=torch.bool, device=first.device)
        cur = 0
        for t in tensor_list:
            if video:
                k = t.shape[1]
                out[cur : cur + k, : t.shape[0], : t.shape[2], : t.shape[3]].copy_(t.transpose(0, 1))
                mask[cur : cur + k, : t.shape[2], : t.shape[3]] = False
                cur += k
            else:
                out[cur, : t.shape[0], : t.shape[1], : t.shape[2]].copy_(t)
                mask[cur, : t.shape[1], : t.shape[2]] = False
                cur += 1
        return cls(out, mask)

    def __repr__(self):
        return repr(self.tensors)


class LRUCache:
    """Tiny bounded mapping for the per-(durations, inter_idx) device index / target tensors: a real dataset produces a
    new key almost every batch, so an unbounded dict would leak device memory over an epoch; the synthetic bench (and a
    captured HIP graph, which must keep seeing the same tensors) cycles through a handful of keys."""

    def __init__(self, maxsize: int = 32):
        from collections import OrderedDict

        self.maxsize, self._d = maxsize, OrderedDict()

    def get(self, key, default=None):
        if key in self._d:
            self._d.move_to_end(key)
            return self._d[key]
        return default

    def __contains__(self, key):
        return key in self._d

    def __getitem__(self, key):
        self._d.move_to_end(key)
        return self._d[key]

    def __setitem__(self, key, value):
        self._d[key] = value
        self._d.move_to_end(key)
        while len(self._d) > self.maxsize:
            self._d.popitem(last=False)

    def __len__(self):
        return len(self._d)


class FrameSources:
    """The trunk's input frames as an ordered list of sources instead of one concatenated tensor: each part is
    ``(frames (n_src, 3, H, W) fp32 | uint8, index)`` where ``index`` (device int32 tensor or None) picks the source frame
    of every contributed frame.  The slow clip and the fast frames of a step can then be two views of ONE buffer
    (slow = video[::k], datasets/vidstg.py:250-251) and uint8 pixels are normalised on the device
    (td_frames_to_nhwc) - no torch.cat / index copy of hundreds of MB of pixels, a quarter of the host-to-device bytes.
    Quacks like the ``tensors`` field of a NestedTensor where the trunk needs it (shape, device, dtype, to).

    ``valid`` (optional, one entry per part: device int32 ``(n_src, 2)`` or None): the (rows, columns) of every SOURCE frame
    that hold pixels when videos of different sizes were padded to a common H x W as raw uint8; the input kernel writes
    exactly 0 outside, which is what padding the already NORMALISED frames with zeros gives (util/misc.py:158-170)."""

    def __init__(self, parts, valid=None, index_host=None):
        """``index_host`` (optional, one entry per part): the HOST copy of a part's index list (a tuple of ints) when the producer built the
        device index from one - it lets a consumer reason about WHICH frames a part names without a device-to-host copy (TubeDETR proves
        "the slow clip is every k-th fast frame of the same buffer" from it and then skips recomputing those frames)."""
        self.parts = [(t, (i.to(torch.int32).contiguous() if i is not None else None)) for t, i in parts]
        self.index_host = [None] * len(self.parts) if index_host is None else [(tuple(int(v) for v in h) if h is not None else None) for h in index_host]
        assert len(self.index_host) == len(self.parts)
        for (_, i), h in zip(self.parts, self.index_host):
            assert h is None or (i is not None and len(h) == i.numel()), "index_host: the host copy of that part's index list"
        self.valid = [None] * len(self.parts) if valid is None else [(v.to(torch.int32).contiguous() if v is not None else None) for v in valid]
        assert len(self.valid) == len(self.parts)
        for (t, _), v in zip(self.parts, self.valid):
            assert v is None or tuple(v.shape) == (t.shape[0], 2), "valid: (rows, columns) per source frame"
        t0 = self.parts[0][0]
        assert all(t.dim() == 4 and t.shape[1:] == t0.shape[1:] and t.dtype == t0.dtype for t, _ in self.parts), "sources must share C, H, W and dtype"
        assert t0.dtype in (torch.float32, torch.uint8), "frames must be fp32 (normalised) or uint8 pixels"

    @property
    def n_frames(self):
        return sum((i.numel() if i is not None else t.shape[0]) for t, i in self.parts)

    @property
    def shape(self):
        return torch.Size((self.n_frames,) + tuple(self.parts[0][0].shape[1:]))

    @property
    def device(self):
        return self.parts[0][0].device

    @property
    def dtype(self):
        return self.parts[0][0].dtype

    def to(self, device, non_blocking: bool = False):
        return FrameSources([(t.to(device, non_blocking=non_blocking), (i.to(device, non_blocking=non_blocking) if i is not None else None)) for t, i in self.parts],
                            [(v.to(device, non_blocking=non_blocking) if v is not None else None) for v in self.valid], self.index_host)

    def materialize(self) -> torch.Tensor:
        """The concatenated (N, 3, H, W) tensor the sources describe (tests / fallbacks); raw values - the zeroing of
        the padded area described by ``valid`` happens where the pixels are normalised."""
        return torch.cat([(t[i.long()] if i is not None else t) for t, i in self.parts])


def split_into_windows(batch: dict, div_vid: int) -> dict:
    """Windowed inference over long videos (util/misc.py:70-101, ``video_collate_fn(..., div_vid)``): every video of
    ``durations`` is cut into ceil(t / div_vid) consecutive forward windows; captions and video ids are repeated per
    window and the annotated interval is clipped to each window ([-100, -100] where it does not intersect).  Returns the
    updated entries (durations, captions, video_ids, inter_idx); PostProcessSTVG re-assembles windows sharing a video id."""
    import math

    if not div_vid:
        return batch
    out = dict(batch)
    n_fwds = [math.ceil(t / div_vid) for t in batch["durations"]]
    out["durations"] = [min(div_vid, t - i * div_vid) for t, n in zip(batch["durations"], n_fwds) for i in range(n)]
    out["captions"] = [c for c, n in zip(batch["captions"], n_fwds) for _ in range(n)]
    out["video_ids"] = [v for v, n in zip(batch["video_ids"], n_fwds) for _ in range(n)]
    inter = []
    for (start, end), n in zip(batch["inter_idx"], n_fwds):
        for i in range(n):
            lo, hi = max(i * div_vid, start), min((i + 1) * div_vid - 1, end)
            inter.append([-100, -100] if lo > hi else [lo - i * div_vid, hi - i * div_vid])
    out["inter_idx"] = inter
    return out
