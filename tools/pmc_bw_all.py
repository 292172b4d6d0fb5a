"""Achieved HBM bytes per second of EVERY kernel of one training iteration: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE passes (the
directories tools/pmc_to_json.py reads, with their kernel_trace CSVs) -> a table sorted by time.  Bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB
(MI355X_MICROARCH.md, gfx950 correction); durations from the FETCH pass's kernel trace.  A pass that is bound by HBM sits at 4.8-5.5 TB/s
on this pool (torch's copy 5.0-5.5, an elementwise pass 6.0-6.3: profiles/r6_copy_bw_probe.txt); far below = bound by something else.
    python tools/pmc_bw_all.py <fetch dir> <write dir> [min ms]"""
import collections
import csv
import glob
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?([\w:<>, ]+?)\(", name)
    s = (m.group(1) if m else name).strip()
    return s if len(s) <= 90 else s[:87] + "..."


def counter(root, name):
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] == name:
                    k = short(row["Kernel_Name"])
                    tot[k] += float(row["Counter_Value"])
                    cnt[k] += 1
    return tot, cnt


def durations(root):
    tot = collections.defaultdict(float)
    for path in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                tot[short(row["Kernel_Name"])] += (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e6
    return tot


def main(fetch, write, min_ms=0.3):
    f, n = counter(fetch, "FETCH_SIZE")
    w, _ = counter(write, "WRITE_SIZE")
    ms = durations(fetch)
    print("| kernel | launches | ms (under the counter pass) | HBM read GB | HBM written GB | TB/s |")
    print("|---|---:|---:|---:|---:|---:|")
    for k in sorted(ms, key=lambda k: -ms[k]):
        if ms[k] < min_ms:
            continue
        rd, wr = 2 * f.get(k, 0.0) * 1024 / 1e9, w.get(k, 0.0) * 1024 / 1e9
        print(f"| `{k}` | {n.get(k, 0)} | {ms[k]:.2f} | {rd:.2f} | {wr:.2f} | {(rd + wr) / ms[k]:.2f} |")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 0.3)
