"""Vector-pipe accounting of a kernel's basic blocks from the compiler's assembly (hipcc --save-temps, *.s).

On gfx950 the f32-input MFMA (v_mfma_f32_16x16x4_f32: 32 cycles) issues on the SAME vector ALUs as every other v_*
instruction (4 cycles per wave64 instruction; DESIGN_LOG round 4): inside a block, `mfma_frac` = MFMA cycles / (MFMA +
other VALU cycles) is the ceiling the instruction mix alone puts on the MFMA-busy fraction of a wave that owns its SIMD.

    python tools/isa_blocks.py file.s kernel_symbol_substring [min_mfma]
"""
import re
import sys


def main():
    path, sym = sys.argv[1], sys.argv[2]
    min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^\S*%s\S*:" % re.escape(sym), l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    blocks, cur, name = [], [], "entry"
    for l in lines[start + 1:end + 1]:
        t = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            blocks.append((name, cur))
            name, cur = m.group(1), []
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        cur.append(t.split()[0])
    blocks.append((name, cur))
    tot = {"mfma": 0, "valu": 0}
    print(f"{'block':>14} {'mfma':>5} {'valu':>5} {'ds':>4} {'vmem':>4} {'salu':>5} {'wait':>4} {'bar':>3}  mfma_frac")
    for name, ins in blocks:
        mf = sum(1 for i in ins if i.startswith("v_mfma"))
        va = sum(1 for i in ins if i.startswith("v_") and not i.startswith("v_mfma"))
        ds = sum(1 for i in ins if i.startswith("ds_"))
        vm = sum(1 for i in ins if i.startswith(("global_", "buffer_", "flat_", "scratch_")))
        sa = sum(1 for i in ins if i.startswith("s_") and not i.startswith(("s_waitcnt", "s_barrier", "s_nop")))
        wt = sum(1 for i in ins if i.startswith("s_waitcnt"))
        br = sum(1 for i in ins if i.startswith("s_barrier"))
        tot["mfma"] += mf
        tot["valu"] += va
        if mf >= min_mfma:
            print(f"{name:>14} {mf:5d} {va:5d} {ds:4d} {vm:4d} {sa:5d} {wt:4d} {br:3d}  {32 * mf / (32 * mf + 4 * va):.3f}")
    print("static totals:", tot)


if __name__ == "__main__":
    main()
