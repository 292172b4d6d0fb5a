"""rocprofv3 --pmc CSVs -> profiles/pmc_traffic.json (the HBM-traffic numbers bench.py relays as `roofline.traffic`).

PMC counters cannot be read from inside bench.py's timed region, so they are collected in separate profiler passes over the
SAME training iteration and committed together with the commit hash they were taken at:

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python bench.py --steps 1 --warmup 1 --no-prof --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python bench.py --steps 1 --warmup 1 --no-prof --no-cpu-baseline
    python tools/pmc_to_json.py $OUT/fetch $OUT/write --commit $(git rev-parse --short HEAD) > profiles/pmc_traffic.json

(FETCH_SIZE and WRITE_SIZE do not fit one pass: TCC has 4 slots, they cost 3 + 2 — MI355X_MICROARCH.md §rocprofv3 PMC slots.)
Units / corrections, as that guide prescribes for gfx950: the counters are KiB; FETCH_SIZE reports half of the bytes of wide
coalesced reads, so HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  Per family: total bytes / launches over the profiled run.
"""
import argparse
import collections
import csv
import glob
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_digest():
    """Same digest as bench.py's: bench.py marks the relayed traffic as stale when the kernel sources have changed since."""
    h = hashlib.sha1()
    for path in sorted(glob.glob(os.path.join(ROOT, "gif_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "gif_amd", "csrc", "*.h"))
                       + glob.glob(os.path.join(ROOT, "include", "*.h"))):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]

FAMILY_OF = [  # (regex on the kernel name, family key used by bench.py); first match wins
    (r"conv_gather_mfma_glds(_multi)?<_Float16", "conv_gather_mfma_glds_f16"),
    (r"conv_wgrad_mfma<_Float16", "conv_wgrad_mfma_f16"),
    # conv_gather_mfma_glds<T, BM, BN, WM, WN, SCALE, BK, X3, NST>: X3 (2 = f16x2, 1 = bf16x3) is the second-to-last argument, NST the last
    (r"conv_gather_mfma_glds(_multi)?<float,.*, 2, [23]>$", "conv_gather_mfma_glds_h2"),
    (r"conv_wgrad_mfma<float,.*, 2>$|conv_wgrad_h2v2<", "conv_wgrad_mfma_h2"),  # v2 = the cooperative pre-split form (default)
    (r"wino_gemm_h2", "wino_gemm_h2"),
    # (in f16x2 mode the bf16x3 families also contain the guarded twin launches that return at once: their per-launch average is low)
    (r"conv_gather_mfma_glds(_multi)?<float,.*, 1, 2>$", "conv_gather_mfma_glds_x3"),
    (r"conv_wgrad_mfma<float,.*, (1|true)>$", "conv_wgrad_mfma_x3"),
    (r"wino_gemm_x3", "wino_gemm_x3"),
    (r"conv_gather_mfma_glds", "conv_gather_mfma_glds"),
    (r"conv_gather_mfma<", "conv_gather_mfma"),
    (r"wino_gemm_mfma", "wino_gemm_mfma"),
    (r"conv_wgrad_mfma|conv_wgrad_small_mfma", "conv_wgrad_mfma"),
    (r"wino_input_transform|wino_gy_transform", "wino_transforms"),
]


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?([\w:<>, ]+?)\(", name)
    return (m.group(1) if m else name).strip()


def read_counter(root, counter):
    """kernel name -> (sum of the counter over dispatches, dispatches)"""
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    files = glob.glob(root + "/**/*counter_collection.csv", recursive=True)
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {root}")
    for path in files:
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                k = short(row["Kernel_Name"])
                tot[k] += float(row["Counter_Value"])
                cnt[k] += 1
    return tot, cnt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("--commit", default="unknown")
    ap.add_argument("--command", default="python bench.py --steps 1 --warmup 1 --no-prof --no-cpu-baseline")
    a = ap.parse_args()
    fetch, nf = read_counter(a.fetch_dir, "FETCH_SIZE")
    write, nw = read_counter(a.write_dir, "WRITE_SIZE")
    fams = collections.OrderedDict()
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        n = max(nf.get(k, 0), nw.get(k, 0))
        hbm = (2.0 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024.0
        kernels[k] = {"launches": n, "fetch_kib": fetch.get(k, 0.0), "write_kib": write.get(k, 0.0),
                      "hbm_bytes_per_launch": hbm / n if n else None}
        for rx, fam in FAMILY_OF:
            if re.search(rx, k):
                f = fams.setdefault(fam, {"kernels": [], "launches": 0, "hbm_bytes": 0.0})
                f["kernels"].append(k)
                f["launches"] += n
                f["hbm_bytes"] += hbm
                break
    for f in fams.values():
        f["hbm_bytes_per_launch"] = f["hbm_bytes"] / f["launches"] if f["launches"] else None
    top = sorted(kernels.items(), key=lambda kv: -(kv[1]["hbm_bytes_per_launch"] or 0) * kv[1]["launches"])[:25]
    json.dump({"commit": a.commit, "kernel_source_digest": kernel_source_digest(), "command": a.command,
               "units": "HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (counters in KiB; gfx950 FETCH_SIZE halving corrected)",
               "families": fams, "top_kernels_by_traffic": dict(top)}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
