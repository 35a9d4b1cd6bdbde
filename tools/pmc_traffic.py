"""Per-kernel-family HBM traffic from rocprofv3 PMC passes (MI355X_MICROARCH.md, HBM section: FETCH_SIZE and
WRITE_SIZE cannot share a pass; on gfx950 FETCH_SIZE counts a 16-byte/lane streaming read at half its bytes -> x2;
both are in KiB).  Input: the two per-kernel aggregates written by the collection command in profiles/README.md
(kernel,counter,dispatches,sum).  Output: profiles/<round>_pmc_traffic.json, read by bench.py for `roofline.traffic`.

usage: python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE.csv gpurun_out/pmc_WRITE_SIZE.csv profiles/r01_pmc_traffic.json"""
import csv, json, re, sys


def family(name):
    m = re.search(r"td::conv_gemm_kernel<([^,]+), (\d+), (\d+), (\d+), (true|false)", name)
    if m:
        return f"td::conv_gemm_kernel<{m.group(1)}, {m.group(2)}, {m.group(3)}, *, *>"  # stages and pointwise flag merged
    if "td::pw_resident_kernel" in name or "td::pw_resident2_kernel" in name:
        return "td::pw_resident_kernel<*>"
    m = re.search(r"td::conv_gemm_big(?:8n?)?_kernel<(?:\d+, )?(true|false)>", name)  # lock-step and phased (256 x 256, 256 x 128) instances: one family
    if m:  # the 256-row tile kernel on spatial (3x3, MFMA-bound) and on pointwise K >= 512 (HBM-bound) layers
        return f"td::conv_gemm_big_kernel<*, {m.group(1)}>"
    if "td::stem_pool_kernel" in name or re.search(r"td::bottleneck_(fused|resident|resident3|first3)_kernel", name):
        return "td::stem_pool_kernel<*> + td::bottleneck_fused_kernel<*>"
    if "td::cross_q1_" in name:
        return "td::cross_q1_*_kernel"
    if "td::conv_wgrad_wide_batch_kernel" in name or "td::conv_wgrad_batch_kernel" in name:
        return "td::conv_wgrad_*batch_kernel"
    m = re.search(r"td::(conv_wgrad(?:_batch)?_kernel)<([^,>]+)", name)
    if m:
        return f"td::{m.group(1)}<{m.group(2)}>"
    m = re.search(r"(?<![a-z_])(td::[a-z0-9_]+)", name)
    return m.group(1) if m else None


def load(path):
    out = {}
    for r in csv.DictReader(open(path)):
        f = family(r["kernel"])
        if f is None:
            continue
        a = out.setdefault(f, [0, 0.0])
        a[0] += int(r["dispatches"])
        a[1] += float(r["sum"])
    return out


def main():
    fetch, write, dst = sys.argv[1:4]
    F, W = load(fetch), load(write)
    res = {"_doc": "bytes per launch = counter sum (KiB) * 1024 / dispatches; fetch doubled (gfx950 16-B/lane streaming-read correction); "
                   "Infinity-Cache hits are counted by these memory-side counters",
           "_command": "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace -- python bench.py --no-graph --steps 1 --warmup 1 --cpu-frames 0 --roofline-steps 0",
           "_steps": 2}  # warm-up + timed step: `dispatches` / _steps = launches per step
    for f in sorted(set(F) | set(W)):
        nf, sf = F.get(f, [0, 0.0])
        nw, sw = W.get(f, [0, 0.0])
        res[f] = {"dispatches": nf or nw,
                  "fetch_bytes_per_launch": round(sf * 1024 * 2 / nf) if nf else None,
                  "write_bytes_per_launch": round(sw * 1024 / nw) if nw else None}
    json.dump(res, open(dst, "w"), indent=1)
    for f, v in res.items():
        if isinstance(v, dict) and v.get("fetch_bytes_per_launch"):
            print(f"{f:70s} n={v['dispatches']:5d} fetch {v['fetch_bytes_per_launch']/1e6:9.2f} MB  write {(v['write_bytes_per_launch'] or 0)/1e6:9.2f} MB per launch")


if __name__ == "__main__":
    main()
