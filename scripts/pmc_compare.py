#!/usr/bin/env python3
"""Side-by-side table of two scripts/pmc_summary.py outputs (before / after), per kernel: the counters VERDICT r5 asked for.
usage: pmc_compare.py before.txt after.txt"""
import re
import sys

WANT = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "FETCH_SIZE", "WRITE_SIZE"]


def load(path):
    out, cur = {}, None
    for line in open(path):
        if not line.startswith(" "):
            cur = re.sub(r"<([0-9]), (true|false)(, (true|false))*>", lambda m: "<" + m.group(1) + ">", line.strip().replace("bv::", ""))
            cur = cur.replace(", RangeView", "").replace(", bv::RangeView", "")
            out.setdefault(cur, {})
        else:
            m = re.match(r"\s+(\S+)\s+mean/dispatch\s+([0-9.]+)\s+dispatches\s+(\d+)", line)
            if m and cur:
                d = out[cur].setdefault(m.group(1), [0.0, 0])
                d[0] += float(m.group(2)) * int(m.group(3)); d[1] = max(d[1], int(m.group(3)))
    return out


def main():
    a, b = load(sys.argv[1]), load(sys.argv[2])
    def scans(t):  # a kernel that is launched once per scan
        for pref in ("k_parse_list", "k_parse_tile", "k_scan_sums"):
            c = max((v.get("SQ_INSTS_VALU", [0, 0])[1] for k, v in t.items() if k.startswith(pref)), default=0)
            if c:
                return c
        return 1
    scans_a, scans_b = scans(a), scans(b)
    print("per SCAN (sum over the kernel's launches of one scan), millions; FETCH / WRITE in MB as the counters report them (KB x 1024, no gfx950 correction); before -> after")
    print("%-34s " % "kernel" + " ".join("%19s" % w.replace("SQ_", "").replace("INSTS_", "") for w in WANT))
    tot = {w: [0.0, 0.0] for w in WANT}
    for k in sorted(set(a) | set(b)):
        row = []
        for w in WANT:
            va = a.get(k, {}).get(w, [0.0, 0])[0] / scans_a
            vb = b.get(k, {}).get(w, [0.0, 0])[0] / scans_b
            sc = 1024.0 / 1e6 if w.endswith("_SIZE") else 1e-6
            tot[w][0] += va * sc; tot[w][1] += vb * sc
            row.append("%8.1f ->%8.1f" % (va * sc, vb * sc))
        if any(a.get(k, {}).get(w, [0])[0] > 2e5 or b.get(k, {}).get(w, [0])[0] > 2e5 for w in WANT[:2]):
            print("%-34s " % k[:34] + " ".join(row))
    print("%-34s " % "ALL KERNELS" + " ".join("%8.1f ->%8.1f" % (tot[w][0], tot[w][1]) for w in WANT))


if __name__ == "__main__":
    main()
