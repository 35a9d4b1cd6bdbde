"""Kernel families of the roofline report: ONE place that says which launched kernels (names as rocprofv3 prints them) form a
family, shared by bench.py (in-library HIP-event timing per TD_PROF id) and by the PMC aggregation (tools/pmc_collect.py,
tools/pmc_traffic.py).  A family's key is the list of its kernels' name stems - every stem is a substring of the kernel names in the
committed kernel-stats CSV (profiles/<round>_bench_cfg3x16_kernel_stats.csv); the PMC JSONs add the exact names under "kernels"."""
import re

# TD_PROF family id (include/tubedetr_hip.h) -> key
FAMILY_OF_PROF_ID = {
    0: "td::conv_gemm_kernel<unsigned short, 128, 128, ...>",
    1: "td::conv_gemm_kernel<unsigned short, 128, 64, ...>",
    3: "td::conv_gemm_kernel<unsigned short, 64, 128, ...>",
    2: "td::conv_wgrad_wide_batch_kernel + td::conv_wgrad_batch_kernel",
    4: "td::pw_resident2_kernel",
    5: "td::conv_gemm_big8_kernel<true, ...> + td::conv_gemm_big8n_kernel<true>",     # 256-row tiles, spatial (3x3 / strided) layers: MFMA-bound
    6: "td::conv_gemm_big8_kernel<false, ...> + td::conv_gemm_big8n_kernel<false>",   # 256-row tiles, pointwise K >= 512: HBM / MFMA co-limited
    7: "td::stem_pool_kernel + td::bottleneck_first3_kernel + td::bottleneck_resident3_kernel",
    8: "td::cross_q1_fwd_mfma_kernel + td::cross_q1_bwd_mfma_kernel + td::cross_q1_dmem_kernel",
    9: "td::conv_wgrad_kernel<unsigned short, ...>",
    10: "td::pw_chain2_kernel_w4",  # conv3 + identity of a layer3 block chained with the next block's conv1: HBM-bound, the block output is not read back
}


def prof_key(fam_id: int, fp32: bool = False) -> str:
    k = FAMILY_OF_PROF_ID[fam_id]
    return k.replace("unsigned short", "float") if fp32 else k


def family(name: str):
    """kernel name (rocprofv3 Kernel_Name) -> family key, or the bare td:: kernel name for kernels outside the reported families."""
    m = re.search(r"td::conv_gemm_kernel<([^,]+), (\d+), (\d+),", name)
    if m:
        return f"td::conv_gemm_kernel<{m.group(1)}, {m.group(2)}, {m.group(3)}, ...>"  # pipeline depths, pointwise / tap-uniform / two-source instances merged
    if "td::pw_resident2_kernel" in name:
        return FAMILY_OF_PROF_ID[4]
    if "td::pw_chain2_kernel" in name:
        return FAMILY_OF_PROF_ID[10]
    m = re.search(r"td::conv_gemm_big8n?_kernel<(true|false)", name)
    if m:
        return FAMILY_OF_PROF_ID[5 if m.group(1) == "true" else 6]
    if re.search(r"td::(stem_pool|bottleneck_first3|bottleneck_resident3)_kernel", name):
        return FAMILY_OF_PROF_ID[7]
    if re.search(r"td::cross_q1_(fwd|bwd)(_mfma)?_kernel|td::cross_q1_dmem_kernel", name):
        return FAMILY_OF_PROF_ID[8]
    if "td::conv_wgrad_wide_batch_kernel" in name or "td::conv_wgrad_batch_kernel" in name:
        return FAMILY_OF_PROF_ID[2]
    m = re.search(r"td::conv_wgrad_kernel<([^,>]+)", name)
    if m:
        return f"td::conv_wgrad_kernel<{m.group(1)}, ...>"
    m = re.search(r"(?<![a-z_])(td::[a-z0-9_]+)", name)
    return m.group(1) if m else None
