"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace into a per-kernel stats table
(the `--stats` CSV equivalent): calls, total / average / min / max duration, share of GPU time.
Usage: python tools/rocpd_stats.py gpurun_out/prof/x_results.db [> profiles/name.md]"""
import os
import re
import subprocess
import sqlite3
import sys


_FILT = "/usr/bin/c++filt"


def short(name):
    if name.startswith("_Z") and os.path.exists(_FILT):  # rocprofv3 leaves _Float16 instantiations (DF16_) mangled; binutils knows Dh
        name = subprocess.run([_FILT, name.replace("DF16_", "Dh")], capture_output=True, text=True).stdout.strip() or name
        name = name.replace("__fp16", "f16").replace("_Float16", "f16")
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:<>, ]+?)\(", name)
    s = m.group(1) if m else name
    return s if len(s) <= 110 else s[:107] + "..."


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {namecol}, start, end from kernels").fetchall()
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(short(n), [0, 0.0, 1e30, 0.0])
        d = (e - s) / 1e3  # ns -> us
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    print(f"| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| `{n}` | {a[0]} | {a[1] / 1e3:.2f} | {a[1] / a[0]:.1f} | {a[2]:.1f} | {a[3]:.1f} | {100 * a[1] / total:.1f} |")
    print(f"\ntotal kernel time {total / 1e3:.1f} ms over {len(rows)} dispatches")
    # how much of the trace's span the GPU had no kernel running (launch gaps, host stalls): union of the busy intervals against the span of
    # the LAST 60 % of the dispatches (warm-up, allocator growth and the profiler's own start-up are in the first part)
    iv = sorted((s, e) for _, s, e in rows)
    iv = iv[int(len(iv) * 0.4):]
    if iv:
        busy, gaps, cur_s, cur_e = 0, [], iv[0][0], iv[0][1]
        for s, e in iv[1:]:
            if s > cur_e:
                busy += cur_e - cur_s
                gaps.append(s - cur_e)
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        busy += cur_e - cur_s
        span = cur_e - iv[0][0]
        big = [g for g in gaps if g > 20000]
        print(f"idle inside the last 60 % of the trace: {(span - busy) / 1e6:.1f} ms of {span / 1e6:.1f} ms ({100.0 * (span - busy) / span:.1f} %), "
              f"{len(gaps)} gaps, median {sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0:.1f} us, {len(big)} gaps > 20 us totalling {sum(big) / 1e6:.1f} ms")


if __name__ == "__main__":
    main(sys.argv[1])
