"""The LDS bank model behind the layouts of the fused layer1 kernels (tools/lds_bank_model.py; no GPU): the round-6 layouts must be
conflict-free on every ds_read_b128 pattern the kernels issue, and the model must keep reproducing what rocprofv3 measured for the
round-5 layouts (three-way fragment reads of conv2, two-way everything else) - that agreement is what the new layouts were chosen by."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import lds_bank_model as m  # noqa: E402


def test_a_conflict_free_wave_read_costs_one_cycle_per_lane_group():
    assert m.cycles_read_b128([16 * l for l in range(64)]) == 4            # 64 consecutive 16-byte slots
    assert m.cycles_read_b128([256 * l for l in range(64)]) == 64          # every lane on the same four banks
    assert m.cycles_read_b128([0] * 64) == 4                               # identical addresses broadcast


def test_round_6_layouts_are_conflict_free_on_every_fragment_read():
    r = m.resident3(160, "8x2")
    assert all(v == 4 for k, v in r.items() if "reads" in k), r
    assert all(v == 4 for v in m.first3(160).values())


def test_round_5_layouts_reproduce_the_measured_conflict_ratio():
    r = m.resident3(144, "2x8")
    assert r["conv2 fragment reads (36 per wavefront and tile)"] == 12 and r["conv3 fragment reads of h2 (4)"] == 8
    # per wavefront and tile: 32 conv1 + 36 conv2 + 4 conv3 + 8 identity reads, 6 result stores, 8 staging stores (two-way) and 4 staging reads
    cyc = 32 * 4 + 36 * 12 + 4 * 8 + 8 * 8 + 6 * 8 + 8 * 8 + 4 * 4
    free = (32 + 36 + 4 + 8 + 6 + 8 + 4) * 4
    assert abs((cyc - free) / cyc - 0.50) < 0.03   # SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.50 (profiles/r06_pmc_LDS_per_kernel_before_layer1_rework.csv)
    assert all(v == 8 for v in m.first3(144).values())
