#!/usr/bin/env python3
"""Per-kernel register / scratch / spill figures of the gfx950 code object inside libmpmvs_hip.so.

Reads the metadata notes the compiler wrote (.vgpr_count, .vgpr_spill_count, .private_segment_fixed_size, ...), i.e. what
the hardware launch descriptor is built from -- not an estimate.  Usage: tools/kernel_resources.py [--filter k_update] [--csv]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def extract_code_object(so, workdir):
    fat = os.path.join(workdir, "fatbin")
    co = os.path.join(workdir, "gfx950.co")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, fat])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"])
    return co


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return out.splitlines()


def kernel_table(co):
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    rows, cur = [], None
    for line in notes.splitlines():
        m = re.match(r"\s+(?:- )?\.(\w+):\s+(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip().strip("'")
        if line.lstrip().startswith("- .") and k in ("agpr_count", "args"):
            if cur:
                rows.append(cur)
            cur = {}
        if cur is not None and k in ("name", "vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
                                     "private_segment_fixed_size", "group_segment_fixed_size", "max_flat_workgroup_size"):
            cur[k] = v
    if cur:
        rows.append(cur)
    rows = [r for r in rows if "name" in r and "vgpr_count" in r]
    for r, d in zip(rows, demangle([r["name"] for r in rows])):
        r["demangled"] = re.sub(r"^void ", "", d).split("(")[0]
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--so", default=os.path.join(ROOT, "mp-mvs_amd", "csrc", "libmpmvs_hip.so"))
    ap.add_argument("--filter", default="")
    ap.add_argument("--csv", action="store_true")
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as wd:
        rows = kernel_table(extract_code_object(args.so, wd))
    rows = [r for r in rows if args.filter in r["demangled"]]
    rows.sort(key=lambda r: r["demangled"])
    cols = ["vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size"]
    if args.csv:
        print("kernel," + ",".join(cols))
        for r in rows:
            print('"%s",' % r["demangled"] + ",".join(r.get(c, "") for c in cols))
        return
    print(f"{'kernel':58s} vgpr agpr sgpr vspill sspill scratch  lds")
    for r in rows:
        print(f"{r['demangled'][:58]:58s} " + " ".join(f"{r.get(c, ''):>5s}" for c in cols))


if __name__ == "__main__":
    sys.exit(main())
