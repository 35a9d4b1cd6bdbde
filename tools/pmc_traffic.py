"""Per-kernel-family HBM traffic from rocprofv3 PMC passes (MI355X_MICROARCH.md, HBM section: FETCH_SIZE and
WRITE_SIZE cannot share a pass; on gfx950 FETCH_SIZE counts a 16-byte/lane streaming read at half its bytes -> x2;
both are in KiB).  Input: the two per-kernel aggregates written by the collection command in profiles/README.md
(kernel,counter,dispatches,sum).  Output: profiles/<round>_pmc_traffic.json, read by bench.py for `roofline.traffic`.

usage: python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE.csv gpurun_out/pmc_WRITE_SIZE.csv profiles/r01_pmc_traffic.json"""
import csv, json, re, sys


import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from kernel_families import family  # noqa: E402  (one definition shared with bench.py)


NAMES = {}  # family -> exact kernel names seen (as rocprofv3 prints them: the names of the committed kernel-stats CSV)


def load(path):
    out = {}
    for r in csv.DictReader(open(path)):
        f = family(r["kernel"])
        if f is None:
            continue
        a = out.setdefault(f, [0, 0.0])
        a[0] += int(r["dispatches"])
        a[1] += float(r["sum"])
        NAMES.setdefault(f, set()).add(r["kernel"])
    return out


def main():
    fetch, write, dst = sys.argv[1:4]
    F, W = load(fetch), load(write)
    res = {"_doc": "bytes per launch = counter sum (KiB) * 1024 / dispatches; fetch doubled (gfx950 16-B/lane streaming-read correction); "
                   "Infinity-Cache hits are counted by these memory-side counters",
           "_command": "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace -- python bench.py --no-graph --steps 1 --warmup 1 --cpu-frames 0 --roofline-steps 0 --dedupe-steps 0",
           "_steps": 2}  # warm-up + timed step: `dispatches` / _steps = launches per step
    for f in sorted(set(F) | set(W)):
        nf, sf = F.get(f, [0, 0.0])
        nw, sw = W.get(f, [0, 0.0])
        res[f] = {"kernels": sorted(NAMES.get(f, [])), "dispatches": nf or nw,
                  "fetch_bytes_per_launch": round(sf * 1024 * 2 / nf) if nf else None,
                  "write_bytes_per_launch": round(sw * 1024 / nw) if nw else None}
    json.dump(res, open(dst, "w"), indent=1)
    for f, v in res.items():
        if isinstance(v, dict) and v.get("fetch_bytes_per_launch"):
            print(f"{f:70s} n={v['dispatches']:5d} fetch {v['fetch_bytes_per_launch']/1e6:9.2f} MB  write {(v['write_bytes_per_launch'] or 0)/1e6:9.2f} MB per launch")


if __name__ == "__main__":
    main()
