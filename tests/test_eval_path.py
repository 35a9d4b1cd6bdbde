"""Evaluation-side path against golden vectors produced by the reference itself
(tests/golden/eval_path.npz, oracle/gen_golden_eval.py): PostProcessSTVG incl. multi-window ensembling, PostProcess, the
windowed collation bookkeeping, the learning-rate schedules.  CPU tests pin the oracle restatement and the host logic;
the GPU tests run the HIP post-processor."""
import json
import os
import types

import numpy as np
import pytest
import torch

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "eval_path.npz"))


def _sted_case(name):
    return (torch.from_numpy(GOLD[f"sted.{name}.steds"]), torch.from_numpy(GOLD[f"sted.{name}.time_mask"]), [str(v) for v in GOLD[f"sted.{name}.video_ids"]],
            json.loads(str(GOLD[f"sted.{name}.frames_id"])), GOLD[f"sted.{name}.result"])


@pytest.mark.parametrize("name", ["single", "windows"])
def test_oracle_post_process_matches_reference(name):
    from oracle.eval_oracle import post_process_stvg

    steds, tm, vids, fids, want = _sted_case(name)
    assert np.array_equal(np.array(post_process_stvg(steds, fids, vids, tm)), want)


def test_box_post_process_matches_reference():
    from tubedetr_amd.models.postprocessors import PostProcess, build_postprocessors

    res = PostProcess()({"pred_boxes": torch.from_numpy(GOLD["bbox.boxes"])}, torch.from_numpy(GOLD["bbox.sizes"]))
    assert np.allclose(torch.stack([r["boxes"] for r in res]).numpy(), GOLD["bbox.result"], rtol=0, atol=1e-5)
    assert set(build_postprocessors(None, "vidstg")) == {"bbox", "vidstg"} and set(build_postprocessors(None, "x")) == {"bbox"}


def test_window_split_matches_reference_collate():
    from tubedetr_amd.util.misc import split_into_windows

    if "windows.error" in GOLD:
        pytest.skip(str(GOLD["windows.error"]))
    durations = [int(x) for x in GOLD["windows.durations_in"]]
    batch = {"durations": durations, "captions": [f"c{i}" for i in range(len(durations))], "video_ids": [f"v{i}" for i in range(len(durations))],
             "inter_idx": [list(map(int, r)) for r in GOLD["windows.inter_in"]]}
    out = split_into_windows(batch, int(GOLD["windows.div"]))
    assert out["durations"] == [int(x) for x in GOLD["windows.durations"]]
    assert out["inter_idx"] == [list(map(int, r)) for r in GOLD["windows.inter_idx"]]
    assert out["video_ids"] == [str(x) for x in GOLD["windows.video_ids"]] and out["captions"] == [str(x) for x in GOLD["windows.captions"]]
    assert split_into_windows(batch, 0) is batch


def test_learning_rate_schedules_match_reference():
    from tubedetr_amd.optim import adjust_learning_rate

    names = ["step", "multistep", "linear_with_warmup", "all_linear_with_warmup"]
    for row in GOLD["lr.rows"]:
        a = types.SimpleNamespace(fraction_warmup_steps=0.01, schedule=names[int(row[0])], lr_drop=10, epochs=120, lr=5e-5, lr_backbone=1e-5, text_encoder_lr=5e-5)
        opt = types.SimpleNamespace(param_groups=[{"lr": 0.0}, {"lr": 0.0}, {"lr": 0.0}])
        adjust_learning_rate(opt, int(row[1]), int(row[2]), num_training_steps=100000, args=a)
        assert np.allclose([g["lr"] for g in opt.param_groups], row[3:], rtol=1e-12, atol=0), row


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["single", "windows"])
def test_hip_post_process_matches_reference(name):
    from tubedetr_amd.models.postprocessors import PostProcessSTVG

    steds, tm, vids, fids, want = _sted_case(name)
    dev = torch.device("cuda:0")
    got = PostProcessSTVG()({"pred_sted": steds.to(dev)}, frames_id=fids, video_ids=vids, time_mask=tm.to(dev))
    assert np.array_equal(np.array(got), want)


@pytest.mark.gpu
def test_hip_post_process_matches_oracle_on_long_videos():
    from oracle.eval_oracle import post_process_stvg
    from tubedetr_amd.models.postprocessors import PostProcessSTVG

    g = torch.Generator().manual_seed(9)
    B, T = 6, 200
    steds = torch.randn(B, T, 2, generator=g) * 4
    tm = torch.ones(B, T, dtype=torch.bool)
    tm[1, 150:] = False
    tm[5, 60:] = False
    vids = ["a", "a", "b", "c", "c", "c"]
    fids = [list(range(400)), list(range(10, 210)), list(range(0, 1200, 2))]
    dev = torch.device("cuda:0")
    got = PostProcessSTVG()({"pred_sted": steds.to(dev)}, frames_id=fids, video_ids=vids, time_mask=tm.to(dev))
    assert got == post_process_stvg(steds, fids, vids, tm)
    uniq = [f"v{i}" for i in range(B)]
    f2 = [list(range(T))] * B
    assert PostProcessSTVG()({"pred_sted": steds.to(dev)}, frames_id=f2, video_ids=uniq, time_mask=tm.to(dev)) == post_process_stvg(steds, f2, uniq, tm)
