#!/usr/bin/env python3
"""where a kernel's VGPR spills and reloads sit relative to its gathers (from hipcc -S output):
   tools/spills.py build/all.s 'k_update<false, true, 8, false>'
   (build/all.s: hipcc <flags of csrc/Makefile> --cuda-device-only -S -o build/all.s mp-mvs_amd/csrc/mpmvs_api.hip)"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
want = sys.argv[2]
for f in re.split(r'\n\s*\.globl\s+', txt):
    name = f.split('\n', 1)[0].strip()
    if 'k_' not in name:
        continue
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    if want not in dem:
        continue
    lab, gathers = None, 0
    print(dem)
    for i, l in enumerate(f.split('\n')):
        m = re.match(r'^(\.LBB\S+):', l)
        if m:
            lab = m.group(1)
        if 'buffer_load_dwordx' in l and ('offen' in l or 'idxen' in l):
            gathers += 1
        if ('Folded Spill' in l or 'Folded Reload' in l) and 'scratch' in l:
            print(f"  line {i:6d} {lab:14s} gathers so far {gathers:3d}  {l.strip()[:100]}")
