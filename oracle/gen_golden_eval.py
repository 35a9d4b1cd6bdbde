"""Generate tests/golden/eval_path.npz by running the REFERENCE's own evaluation-side code (/root/reference, read-only) on
CPU: PostProcessSTVG / PostProcess (models/postprocessors.py:13-107), the windowed collation of video_collate_fn
(util/misc.py:40-101) and update_ema / adjust_learning_rate (util/optim.py).  TEST INFRASTRUCTURE ONLY; runs only in the
build container.  Usage: python -m oracle.gen_golden_eval   (from the repo root)

The fixture stores inputs AND the reference's outputs (all small), so the tests need neither the reference nor this
script on the GPU box."""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np
import torch

from oracle.gen_golden import OUT, REF, install_stubs


def eval_cases():
    """Deterministic inputs of the post-processor cases: (name, steds [B,T,2], time_mask [B,T], video_ids, frames_id)."""
    g = torch.Generator().manual_seed(123)
    cases = []
    # single window per video
    B, T = 3, 17
    steds = torch.randn(B, T, 2, generator=g) * 3
    tm = torch.ones(B, T, dtype=torch.bool)
    tm[1, 12:] = False
    cases.append(("single", steds, tm, ["v0", "v1", "v2"], [list(range(5, 5 + T)), list(range(0, 2 * 12, 2)), list(range(100, 100 + T))]))
    # two videos cut into 3 + 2 windows of 10 frames (the last ones shorter): ensembled by video id
    B, T = 5, 10
    steds = torch.randn(B, T, 2, generator=g) * 3
    tm = torch.ones(B, T, dtype=torch.bool)
    tm[2, 7:] = False
    tm[4, 4:] = False
    cases.append(("windows", steds, tm, ["a", "a", "a", "b", "b"], [list(range(0, 27)), list(range(0, 27)), list(range(0, 27)), list(range(3, 3 + 14)), list(range(3, 3 + 14))]))
    return cases


def main():
    install_stubs()
    sys.path.insert(0, REF)
    for name in ("torch.utils.tensorboard", "ffmpeg", "cv2"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    from models.postprocessors import PostProcess, PostProcessSTVG  # the reference's own classes
    from util.optim import adjust_learning_rate

    out = {}
    pp = PostProcessSTVG()
    for name, steds, tm, vids, fids in eval_cases():
        frames_id = fids if name == "single" else [fids[0], fids[3]]  # one frame-id list per distinct video in the ensembled case
        res = pp({"pred_sted": steds}, frames_id=frames_id, video_ids=vids, time_mask=tm)
        out[f"sted.{name}.steds"] = steds.numpy()
        out[f"sted.{name}.time_mask"] = tm.numpy()
        out[f"sted.{name}.video_ids"] = np.array(vids)
        out[f"sted.{name}.frames_id"] = np.array(json.dumps(frames_id))
        out[f"sted.{name}.result"] = np.array(res, dtype=np.float64)
    g = torch.Generator().manual_seed(5)
    boxes = torch.rand(6, 4, generator=g)
    sizes = torch.tensor([[480, 640]] * 3 + [[352, 352]] * 3, dtype=torch.float32)
    res = PostProcess()({"pred_boxes": boxes}, sizes)
    out["bbox.boxes"], out["bbox.sizes"], out["bbox.result"] = boxes.numpy(), sizes.numpy(), torch.stack([r["boxes"] for r in res]).numpy()

    # windowed collation: only the list bookkeeping of video_collate_fn (util/misc.py:70-101) - run through the reference's function
    # with one-frame dummy clips so that NestedTensor.from_tensor_list has something to pad
    from util.misc import video_collate_fn

    durations, div = [23, 10, 31], 10
    batch = [(torch.zeros(3, d, 4, 4), [{} for _ in range(d)], {"caption": f"c{i}", "video_id": f"v{i}", "frames_id": list(range(d)), "inter_idx": inter})
             for i, (d, inter) in enumerate(zip(durations, [[5, 17], [0, 9], [12, 30]]))]
    try:
        fb = video_collate_fn(False, div, batch)
        out["windows.durations_in"] = np.array(durations)
        out["windows.inter_in"] = np.array([[5, 17], [0, 9], [12, 30]])
        out["windows.div"] = np.array(div)
        out["windows.durations"] = np.array(fb["durations"])
        out["windows.inter_idx"] = np.array(fb["inter_idx"])
        out["windows.video_ids"] = np.array(fb["video_ids"])
        out["windows.captions"] = np.array(fb["captions"])
    except Exception as e:  # signature drift: record why, the test then skips this part
        out["windows.error"] = np.array(repr(e))

    # learning-rate schedule (util/optim.py:28-95) on a stub optimizer with the three groups
    class Opt:
        def __init__(self):
            self.param_groups = [{"lr": 0.0}, {"lr": 0.0}, {"lr": 0.0}]

    rows = []
    for schedule in ("step", "multistep", "linear_with_warmup", "all_linear_with_warmup"):
        a = types.SimpleNamespace(fraction_warmup_steps=0.01, schedule=schedule, lr_drop=10, epochs=120, lr=5e-5, lr_backbone=1e-5, text_encoder_lr=5e-5)
        for epoch, step in ((0, 0), (0, 50), (3, 2000), (12, 9000), (70, 60000)):
            o = Opt()
            adjust_learning_rate(o, epoch, step, num_training_steps=100000, args=a)
            rows.append([["step", "multistep", "linear_with_warmup", "all_linear_with_warmup"].index(schedule), epoch, step] + [g_["lr"] for g_ in o.param_groups])
    out["lr.rows"] = np.array(rows, dtype=np.float64)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "eval_path.npz"), **out)
    print("wrote", os.path.join(OUT, "eval_path.npz"), sorted(out))


if __name__ == "__main__":
    main()
