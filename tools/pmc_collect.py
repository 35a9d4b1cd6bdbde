"""rocprofv3 PMC output -> per-kernel sums, and the round's MFMA-utilisation table.

  aggregate:  python tools/pmc_collect.py agg <dir or *_counter_collection.csv> out.csv
              per kernel name and counter: dispatches, sum of Counter_Value, summed duration (ns)
              (the format tools/pmc_traffic.py reads: kernel,counter,dispatches,sum[,duration_ns])
  mfma:       python tools/pmc_collect.py mfma <agg.csv> profiles/r02_pmc_mfma.json
              needs SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CU_CYCLES and GRBM_GUI_ACTIVE in the aggregate (one rocprofv3 --pmc
              pass: SQ and GRBM counters come from different blocks, MI355X_MICROARCH.md "rocprofv3 PMC slots").
              mfma_util of a kernel family = sum SQ_VALU_MFMA_BUSY_CYCLES / (sum GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs):
              the fraction of SIMD-cycles of the launches' wall time in which a matrix pipe was busy.  rocprofv3 returns
              GRBM_GUI_ACTIVE summed over the 8 XCDs' GRBM instances (measured: sum / kernel duration = 18 .. 20 cycles/ns
              = 8 x ~2.4 GHz for every large kernel, `xcd_sum_check` in the output), hence the / 8
              (SQ_VALU_MFMA_BUSY_CYCLES counts per-SIMD busy cycles summed over the chip; MI355X_MICROARCH.md, per-instruction
              constants).  Families as in bench.py; plus north_star's group "decoder attention + its K/V projections".
"""
import csv
import glob
import json
import os
import re
import sys


import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from kernel_families import family  # noqa: E402  (one definition shared with bench.py)


def agg(src, dst):
    files = [src] if os.path.isfile(src) else sorted(glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True))
    acc = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            key = (r["Kernel_Name"], r["Counter_Name"])
            a = acc.setdefault(key, [0, 0.0, 0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            try:
                a[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            except Exception:
                pass
    with open(dst, "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(["kernel", "counter", "dispatches", "sum", "duration_ns"])
        for (k, c), (n, s, d) in sorted(acc.items()):
            w.writerow([k, c, n, s, d])
    print(f"{len(acc)} (kernel, counter) rows from {len(files)} file(s) -> {dst}")
    # the same sums per (kernel, grid size): launches of ONE kernel instance are told apart by their grid (the hoisted decoder
    # key / value projections among the other launches of the shared GEMM kernel)
    accg = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            key = (r["Kernel_Name"], r.get("Grid_Size", ""), r["Counter_Name"])
            a = accg.setdefault(key, [0, 0.0, 0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            try:
                a[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            except Exception:
                pass
    with open(os.path.splitext(dst)[0] + "_by_grid.csv", "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(["kernel", "grid", "counter", "dispatches", "sum", "duration_ns"])
        for (k, g_, c), (n, s, d) in sorted(accg.items()):
            w.writerow([k, g_, c, n, s, d])


def mfma(src, dst, command=""):
    per, names = {}, {}
    for r in csv.DictReader(open(src)):
        f = family(r["kernel"])
        if f is None:
            continue
        d = per.setdefault(f, {})
        names.setdefault(f, set()).add(r["kernel"])
        a = d.setdefault(r["counter"], [0, 0.0, 0])
        a[0] += int(r["dispatches"])
        a[1] += float(r["sum"])
        a[2] += int(float(r.get("duration_ns") or 0))
    SIMDS = 256 * 4
    XCDS = 8  # GRBM_GUI_ACTIVE arrives summed over the XCDs
    res = {"_doc": "mfma_util = sum SQ_VALU_MFMA_BUSY_CYCLES / (sum GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs) over the family's launches; "
                   "cu_busy = sum SQ_BUSY_CU_CYCLES / (sum GRBM_GUI_ACTIVE / 8 * 256 CUs); gui_cycles_per_ns = sum GRBM_GUI_ACTIVE / "
                   "summed kernel duration (19.2 = 8 XCDs x 2.4 GHz: the check that the counter is the XCD sum)", "_command": command}

    def util(names):
        mf = ga = bc = dur = 0.0
        n = 0
        for f in names:
            c = per.get(f, {})
            mf += c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0, 0.0, 0])[1]
            ga += c.get("GRBM_GUI_ACTIVE", [0, 0.0, 0])[1]
            bc += c.get("SQ_BUSY_CU_CYCLES", [0, 0.0, 0])[1]
            n += c.get("GRBM_GUI_ACTIVE", [0, 0.0, 0])[0]
            dur += c.get("GRBM_GUI_ACTIVE", [0, 0.0, 0])[2]
        if ga <= 0:
            return None
        return {"dispatches": n, "mfma_util": round(mf / (ga / XCDS * SIMDS), 4), "cu_busy": round(bc / (ga / XCDS * 256), 4) if bc else None,
                "mfma_busy_cycles": mf, "gui_active_cycles": ga, "gui_cycles_per_ns": round(ga / dur, 2) if dur else None}

    for f in sorted(per):
        u = util([f])
        if u:
            u["kernels"] = sorted(names.get(f, []))
            res[f] = u
    # north_star: ">= 40 % MFMA utilisation on the space-time decoder attention (incl. its K/V projections)".  The attention
    # kernels are separable by name; the K/V projections run on the shared GEMM kernel, so the group is reported as the
    # attention kernels alone and as attention + the 64x128 GEMM family that carries the projections (an upper bound on
    # the group's launches: the family also holds the encoder / decoder linears).
    att = [f for f in per if "mha_" in f and ("mfma" in f or "lean" in f)]  # MFMA attention kernels (probs-based and lean)
    g = util(att)
    if g:
        res["decoder_attention_group"] = {"attention_kernels": g, "target": 0.40,
                                          "with_gemm_family_64x128": util(att + [f for f in per if "conv_gemm_kernel" in f and ", 64, 128," in f])}
    # The precise group (per (kernel, grid) rows written by `agg` next to its output): the decoder's attention kernels - the
    # probabilities-based MFMA instances; the encoder runs the lean ones, RoBERTa the head-dim-64 VALU ones - plus the launches of
    # the shared GEMM kernel whose grid is that of the hoisted key / value projections (M = b*t*S rows, N = layers * 256).
    by_grid = os.path.splitext(src)[0] + "_by_grid.csv"
    kv_grid = os.environ.get("TD_KV_GRID", "")
    if os.path.exists(by_grid):
        rows = list(csv.DictReader(open(by_grid)))

        def util_rows(sel):
            mf = ga = dur = 0.0
            n = 0
            for r in rows:
                if not sel(r):
                    continue
                v = float(r["sum"])
                if r["counter"] == "SQ_VALU_MFMA_BUSY_CYCLES":
                    mf += v
                elif r["counter"] == "GRBM_GUI_ACTIVE":
                    ga += v
                    n += int(r["dispatches"])
                    dur += float(r.get("duration_ns") or 0)
            if ga <= 0:
                return None
            return {"dispatches": n, "mfma_util": round(mf / (ga / XCDS * SIMDS), 4), "duration_us": round(dur / 1e3, 1)}

        is_att = lambda r: re.search(r"td::mha_(fwd_mfma_kernel<\d+, false>|bwd_dq_mfma_kernel|bwd_dkv_mfma_kernel)", r["kernel"]) is not None
        is_kv = lambda r: bool(kv_grid) and "td::conv_gemm_kernel" in r["kernel"] and r["grid"] == kv_grid
        is_q1 = lambda r: "td::cross_q1_" in r["kernel"]
        res["decoder_attention_group_precise"] = {
            "definition": "decoder temporal self-attention kernels (mha_*_mfma_kernel, probabilities-based instances) + the time-aligned cross-attention: "
                          "its frame core td::cross_q1_*_kernel (key / value projections moved to the query side: the memory rows are not projected; in bf16 "
                          "the score and weighted-sum products and the deferred d(memory) pass are v_mfma_f32_16x16x32_bf16 tiles, HBM-bound by the memory "
                          "rows: a few MFMAs per kilobyte), or - TD_CROSS_Q1=0 - the hoisted key / value "
                          "projection GEMMs (launches of td::conv_gemm_kernel with grid " + (kv_grid or "<TD_KV_GRID unset>") + ")",
            "target": 0.40, "attention_kernels": util_rows(is_att), "cross_q1_frame_core": util_rows(is_q1), "kv_projections": util_rows(is_kv),
            "group": util_rows(lambda r: is_att(r) or is_kv(r) or is_q1(r))}
    json.dump(res, open(dst, "w"), indent=1)
    for f, v in res.items():
        if isinstance(v, dict) and "mfma_util" in v:
            print(f"{f:66s} n={v['dispatches']:5d}  mfma_util {v['mfma_util']:.3f}  cu_busy {v['cu_busy']}")


if __name__ == "__main__":
    if sys.argv[1] == "agg":
        agg(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "mfma":
        mfma(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "")
    else:
        raise SystemExit(__doc__)
