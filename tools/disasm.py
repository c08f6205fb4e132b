#!/usr/bin/env python3
"""Disassemble one kernel of libmpmvs_hip.so (gfx950 code object) and print an instruction budget.

    tools/disasm.py --kernel 'k_eval_ncc<8, true>' [--out build/k.s] [--loop]

Budget: every VALU opcode is priced with the issue cost measured by tools/ubench_valu.hip on the MI355X (cycles per
wave-instruction per SIMD with >= 2 waves resident: 2 for the full-rate class, 4 for the half-rate class, 8 for the
transcendental class).  With --loop only the hottest basic block (the largest one that ends in a backward branch or, for
fully unrolled code, the largest block) is priced."""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from kernel_resources import extract_code_object, LLVM  # noqa: E402

# measured on gfx950 (profiles/r02_ubench_valu.txt)
FULL_RATE = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_add_u32",
             "v_sub_u32", "v_subrev_u32", "v_fmac_f32", "v_mul_legacy_f32", "v_not_b32", "v_add_co_u32", "v_sub_co_u32", "v_addc_co_u32"}
TRANS = {"v_rcp_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_rsq_f32", "v_sin_f32", "v_cos_f32", "v_fma_mixlo_f16", "v_fma_mixhi_f16",
         "v_rcp_iflag_f32"}


def cost(op):
    base = op.split("_e32")[0].split("_e64")[0].split("_sdwa")[0].split("_dpp")[0]
    if base in TRANS:
        return 8
    if base in FULL_RATE and not (op.endswith("_dpp") or op.endswith("_sdwa")):
        return 2
    return 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--so", default=os.path.join(ROOT, "mp-mvs_amd", "csrc", "libmpmvs_hip.so"))
    ap.add_argument("--kernel", required=True, help="substring of the demangled kernel name")
    ap.add_argument("--out", default="")
    ap.add_argument("--top", type=int, default=40)
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as wd:
        co = extract_code_object(args.so, wd)
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", "-C", co], capture_output=True, text=True).stdout
    blocks, cur, name = {}, None, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            name = m.group(1)
            cur = blocks.setdefault(name, [])
            continue
        if cur is not None and line.strip():
            cur.append(line.strip())
    match = [k for k in blocks if args.kernel in k and not k.startswith("L") and ".kd" not in k]
    if not match:
        sys.exit("no kernel matches; have: " + ", ".join(sorted(k for k in blocks if "k_" in k)[:50]))
    kname = match[0]
    # a kernel's code continues through its local labels (L...) until the next function symbol
    names = list(blocks)
    i = names.index(kname)
    body = list(blocks[kname])
    for nxt in names[i + 1:]:
        if re.match(r"^L\d+", nxt) or nxt.startswith(".L") or nxt.startswith("BB"):
            body.append(f"<{nxt}>:")
            body += blocks[nxt]
        else:
            break
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            f.write(f"; {kname}\n" + "\n".join(body) + "\n")
    ops = [ln.split()[0] for ln in body if not ln.startswith("<") and not ln.startswith("//")]
    hist = collections.Counter(ops)
    valu = {k: v for k, v in hist.items() if k.startswith("v_") and not k.startswith("v_cmp") or k.startswith("v_cmp")}
    total_cycles = sum(cost(k) * v for k, v in valu.items())
    print(f"{kname}: {len(ops)} instructions, {sum(valu.values())} VALU = {total_cycles} issue cycles (static count, whole kernel)")
    for cls, nm in ((2, "full rate (2 cyc)"), (4, "half rate (4 cyc)"), (8, "transcendental (8 cyc)")):
        n = sum(v for k, v in valu.items() if cost(k) == cls)
        print(f"  {nm:24s} {n:6d} instructions {n * cls:7d} cycles")
    other = {k: v for k, v in hist.items() if k not in valu}
    print("  non-VALU:", ", ".join(f"{k} {v}" for k, v in sorted(other.items(), key=lambda kv: -kv[1])[:12]))
    print("  top VALU opcodes:")
    for k, v in sorted(valu.items(), key=lambda kv: -kv[1] * cost(kv[0]))[:args.top]:
        print(f"    {k:28s} {v:5d} x {cost(k)} = {v * cost(k):6d}")


if __name__ == "__main__":
    main()
