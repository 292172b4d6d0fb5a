"""pmcA.txt + pmcB.txt (tools/pmc_summary.py output of the two SQ counter passes of tools/probes/pmc_h2.sh) -> the markdown table of
profiles/r*_pmc_h2.md.  Units: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves, SQ_VALU_MFMA_BUSY_CYCLES cycles
summed over the 1024 SIMDs, SQ_LDS_* cycles summed over the 256 CUs, GRBM_GUI_ACTIVE cycles summed over the 8 XCDs.
    python tools/pmc_table.py <dir with pmcA.txt pmcB.txt>"""
import collections
import re
import sys


def read(path):
    out, cur = collections.defaultdict(dict), None
    for line in open(path):
        if line.startswith("## "):
            cur = line[3:].strip()
        else:
            m = re.match(r"\s+(\S+)\s+mean\s+([\d.]+)\s+\(n=(\d+)\)", line)
            if m and cur:
                out[cur][m.group(1)] = float(m.group(2))
                out[cur]["_n"] = int(m.group(3))
    return out


def main(d):
    a, b = read(d + "/pmcA.txt"), read(d + "/pmcB.txt")
    print("| kernel | launches | kernel Mcycles | MFMA busy per SIMD | waves: active | waiting (any) | waiting to issue | waiting for LDS | VALU inst per wave quad-cycle | LDS array busy per CU | LDS bank-conflict cycles per launch | VALU/MFMA co-execution share of MFMA busy |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
    for k in sorted(a):
        x, y = a[k], b.get(k, {})
        cyc = x.get("GRBM_GUI_ACTIVE", 0) / 8
        wc = x.get("SQ_WAVE_CYCLES", 0)
        if cyc < 1e5 or not wc:
            continue
        mf = x.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
        print(f"| `{k}` | {x['_n']} | {cyc / 1e6:.2f} | {mf / (1024 * cyc):.3f} | {x.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2f} | {x.get('SQ_WAIT_ANY', 0) / wc:.2f} | "
              f"{x.get('SQ_WAIT_INST_ANY', 0) / wc:.2f} | {x.get('SQ_WAIT_INST_LDS', 0) / wc:.3f} | {x.get('SQ_INSTS_VALU', 0) / wc:.3f} | "
              f"{y.get('SQ_LDS_IDX_ACTIVE', 0) / (256 * cyc):.3f} | {y.get('SQ_LDS_BANK_CONFLICT', 0):.0f} | {(y.get('SQ_VALU_MFMA_COEXEC_CYCLES', 0) / mf if mf else 0):.3f} |")


if __name__ == "__main__":
    main(sys.argv[1])
