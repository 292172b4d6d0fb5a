"""GIF_PROF_DUMP csv (family,M,N,K,tag,ms,flops per launch; written by gif_prof_read) -> markdown table per launch shape.
Usage: python tools/shape_table.py shapes.csv STEPS > profiles/rN_conv_shapes.md"""
import collections
import csv
import sys


def main(path, steps):
    agg = collections.OrderedDict()
    for row in csv.reader(open(path)):
        fam, M, N, K, tag, ms, fl = int(row[0]), int(row[1]), int(row[2]), int(row[3]), int(row[4]), float(row[5]), float(row[6])
        a = agg.setdefault((fam, M, N, K, tag), [0, 0.0, 0.0])
        a[0] += 1; a[1] += ms; a[2] += fl
    print("| fam | M | N | K | tag | launches/step | ms/step | TFLOP/s (fam 4: TB/s) |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|")
    tot = collections.defaultdict(float)
    for (fam, M, N, K, tag), (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        tot[fam] += ms / steps
        if ms / steps < 0.4:
            continue
        print(f"| {fam} | {M} | {N} | {K} | {tag} | {n / steps:.1f} | {ms / steps:.2f} | {fl / (ms * 1e-3) / 1e12:.1f} |")
    print("\nTotals (ms/step): " + ", ".join(f"family {f}: {t:.1f}" for f, t in sorted(tot.items())))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]))
