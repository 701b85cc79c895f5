"""Per-kernel resource table from a `hipcc -Rpass-analysis=kernel-resource-usage` log (stderr of a -c compile).

usage: python tools/kres.py LOG [filter-regex]
"""
import re
import subprocess
import sys


def parse(path):
    rows, cur = [], None
    for line in open(path, errors="replace"):
        m = re.search(r"remark:\s+([\w \[\]/-]+?):\s+(\S+) \[-Rpass", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k in ("Function Name", "Name"):
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    return rows


def demangle(names):
    out = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"\(anonymous namespace\)::", "", o).split("(")[0] for o in out]


if __name__ == "__main__":
    rows = parse(sys.argv[1])
    flt = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
    names = demangle([r["name"] for r in rows])
    print(f"{'kernel':58s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'scratch':>7s} {'occ':>4s} {'lds':>7s}")
    for r, n in zip(rows, names):
        if flt and not flt.search(n):
            continue
        print(f"{n[:58]:58s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('TotalSGPRs','?'):>5s} "
              f"{r.get('ScratchSize [bytes/lane]','?'):>7s} {r.get('Occupancy [waves/SIMD]','?'):>4s} {r.get('LDS Size [bytes/block]','?'):>7s}")
