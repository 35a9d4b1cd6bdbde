"""Device-side input path of a training step.

The reference collates normalised fp32 clips on the host (``video_collate_fn`` / ``NestedTensor.from_tensor_list``,
util/misc.py:40-178), builds the slow clip as a second tensor ``video[:, ::stride]`` (datasets/vidstg.py:250-251) and
copies both to the GPU (engine.py:55-57): 186 MB of fp32 pixels per 100-frame clip at res 352, 25 % of them twice.
``ClipPipeline`` sends every pixel ONCE as uint8 (47 MB per clip) from page-locked staging buffers on a copy stream -
double-buffered, so the copy of batch i+1 overlaps the step of batch i - and leaves the rest to the trunk's input kernel
(td_frames_to_nhwc): ImageNet normalisation, NCHW -> NHWC, bf16 cast, and the slow / fast split as an index list over
the one device buffer (``util.misc.FrameSources``).  The produced batch dict is what ``harness.forward_step`` consumes.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import torch

from .util.misc import FrameSources


class ClipPipeline:
    def __init__(self, device, stride: int, depth: int = 2):
        self.device = torch.device(device)
        self.stride = int(stride)
        self.copy_stream = torch.cuda.Stream(self.device)
        self.depth = depth
        self._slots: list = [None] * depth  # (pinned video, pinned mask, event of the last H2D out of them)
        self._next = 0

    def _staging(self, n_frames: int, H: int, W: int):
        i = self._next % self.depth
        self._next += 1
        slot = self._slots[i]
        need = n_frames * 3 * H * W
        if slot is None or slot[0].numel() < need or slot[1].numel() < n_frames * H * W:
            slot = [torch.empty(need, dtype=torch.uint8, pin_memory=True), torch.empty(n_frames * H * W, dtype=torch.bool, pin_memory=True), None]
            self._slots[i] = slot
        if slot[2] is not None:
            slot[2].synchronize()  # the copy that last read this staging pair has finished
        return slot

    def stage(self, videos: Sequence[torch.Tensor], input_ids: torch.Tensor, attention_mask: torch.Tensor, target_boxes: torch.Tensor,
              inter_idx: List[List[int]]) -> dict:
        """videos: one uint8 (T_i, 3, H_i, W_i) CPU tensor per video (decoder output order).  Pads to the batch's max H, W
        (mask True = padding, like NestedTensor.from_tensor_list), packs into page-locked memory and starts the
        host-to-device copies on the copy stream.  Returns a ticket for ``collect``."""
        durations = [int(v.shape[0]) for v in videos]
        H, W = max(int(v.shape[2]) for v in videos), max(int(v.shape[3]) for v in videos)
        n = sum(durations)
        vid_pin, mask_pin, _ = slot = self._staging(n, H, W)
        vid = vid_pin[: n * 3 * H * W].view(n, 3, H, W)
        msk = mask_pin[: n * H * W].view(n, H, W)
        off = 0
        ragged = any(v.shape[2] != H or v.shape[3] != W for v in videos)
        if ragged:
            vid.zero_()  # raw 0 is NOT 0 after normalisation: the padded area is zeroed by the input kernel through `valid_hw`
            msk.fill_(True)
        else:
            msk.fill_(False)
        valid_hw = []
        for v in videos:
            assert v.dtype == torch.uint8 and v.dim() == 4 and v.shape[1] == 3, "videos are uint8 (T, 3, H, W)"
            t, _, h, w = v.shape
            vid[off : off + t, :, :h, :w].copy_(v)
            if ragged:
                msk[off : off + t, :h, :w] = False
            valid_hw += [[h, w]] * t
            off += t
        k = self.stride
        slow_idx, base = [], 0
        for d in durations:  # slow clip = every k-th frame of each video (datasets/vidstg.py:250-251)
            slow_idx += [base + j for j in range(0, d, k)]
            base += d
        with torch.cuda.stream(self.copy_stream):
            vid_dev = vid.to(self.device, non_blocking=True)
            msk_dev = msk.to(self.device, non_blocking=True)
            idx_dev = torch.tensor(slow_idx, dtype=torch.int32).pin_memory().to(self.device, non_blocking=True)
            # extent of every frame inside the padded H x W (None for a uniform batch): the reference pads the NORMALISED
            # frames with zeros (NestedTensor.from_tensor_list, util/misc.py:158-170), so padded pixels must be 0 after the
            # device-side normalisation, not (0 - mean) / std
            vhw_dev = torch.tensor(valid_hw, dtype=torch.int32).pin_memory().to(self.device, non_blocking=True) if ragged else None
            ids_dev = input_ids.pin_memory().to(self.device, non_blocking=True)
            att_dev = attention_mask.pin_memory().to(self.device, non_blocking=True)
            box_dev = target_boxes.pin_memory().to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        slot[2] = ev
        return {"event": ev, "video": vid_dev, "mask": msk_dev, "slow_index": idx_dev, "valid_hw": vhw_dev, "durations": durations, "input_ids": ids_dev,
                "attention_mask": att_dev, "target_boxes": box_dev, "inter_idx": [list(x) for x in inter_idx], "n_slow": len(slow_idx), "slow_index_host": tuple(slow_idx)}

    def collect(self, ticket: dict) -> dict:
        """Batch dict for ``harness.forward_step``; the current stream waits for the ticket's copies (no host sync)."""
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ticket["event"])
        for k_ in ("video", "mask", "slow_index", "valid_hw", "input_ids", "attention_mask", "target_boxes"):
            if ticket[k_] is not None:
                ticket[k_].record_stream(cur)  # allocated on the copy stream, consumed on the compute stream
        video, mask, idx, vhw = ticket["video"], ticket["mask"], ticket["slow_index"], ticket["valid_hw"]
        return {
            "frames": FrameSources([(video, idx)], [vhw], [ticket["slow_index_host"]]),   # slow clip: an index list over the same pixels (+ its host copy: the model proves slow = fast[::k] from it)
            "frames_mask": mask[idx.long()],
            "frames_fast": FrameSources([(video, None)], [vhw]) if vhw is not None else video,  # uint8; normalised by the trunk's input kernel
            "fast_mask": mask,
            "durations": ticket["durations"],
            "input_ids": ticket["input_ids"],
            "attention_mask": ticket["attention_mask"],
            "target_boxes": ticket["target_boxes"],
            "inter_idx": ticket["inter_idx"],
        }
