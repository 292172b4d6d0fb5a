"""How much of a rocprofv3 kernel trace (rocpd sqlite) ran with two or more kernels on the GPU at once, and which pairs.
Usage: python tools/rocpd_overlap.py x_results.db"""
import collections
import sqlite3
import sys

sys.path.insert(0, "tools")
from rocpd_stats import short  # noqa: E402


def main(path):
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else "kernel_name"
    rows = sorted(cur.execute(f"select start, end, {namecol} from kernels").fetchall())
    ev = []
    for i, (s, e, n) in enumerate(rows):
        ev.append((s, 1, i))
        ev.append((e, -1, i))
    ev.sort()
    active, last, busy1, busy2 = set(), None, 0, 0
    pairs = collections.Counter()
    for t, d, i in ev:
        if last is not None and active:
            dt = t - last
            busy1 += dt
            if len(active) >= 2:
                busy2 += dt
                names = sorted(short(rows[j][2])[:60] for j in list(active)[:2])
                pairs[tuple(names)] += dt
        if d > 0:
            active.add(i)
        else:
            active.discard(i)
        last = t
    print(f"kernels {len(rows)}; GPU busy {busy1 / 1e6:.2f} ms; with >= 2 kernels in flight {busy2 / 1e6:.2f} ms ({100.0 * busy2 / max(busy1, 1):.1f} %)")
    for k, v in pairs.most_common(12):
        print(f"  {v / 1e6:8.2f} ms  {k[0]}  ||  {k[1]}")


if __name__ == "__main__":
    main(sys.argv[1])
