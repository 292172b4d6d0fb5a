"""Summarise rocprofv3 --pmc CSV output (counter_collection.csv) per kernel: mean of each counter over dispatches."""
import csv
import collections
import glob
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?([\w:<>, ]+?)\(", name)
    return (m.group(1) if m else name)[:70]


def main(root):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                agg[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for kern, ctrs in sorted(agg.items()):
        if not any(s in kern for s in ("conv_", "conv3x3", "blur", "bias_act", "upfirdn", "wino_")):
            continue
        print(f"## {kern}")
        for c, vals in sorted(ctrs.items()):
            print(f"  {c:32s} mean {sum(vals) / len(vals):16.1f}  (n={len(vals)})")


if __name__ == "__main__":
    main(sys.argv[1])
