#!/usr/bin/env python3
"""Instruction statistics per kernel of a csrc/*.hip file (hipcc -S, device only): tools/isa_stats.py deepfm 'wgrad|fwd3ILi7ELi0'"""
import re, subprocess, sys
from collections import Counter
f, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else '.')
out = f'/tmp/{f}.s'
subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value', '-DDT_UNUSED',
                '-x', 'hip', '-S', '--cuda-device-only', f'/root/repo/deeptables_amd/csrc/{f}.hip', '-o', out], check=True,
               stderr=subprocess.DEVNULL)
s = open(out).read().split('\n')
i = 0
while i < len(s):
    m = re.match(r'^(_Z\w+):', s[i])
    if m and re.search(pat, m.group(1)):
        name = m.group(1)
        j = i + 1
        body = []
        while j < len(s) and not s[j].startswith('.Lfunc_end'):
            l = s[j].strip()
            if l and not l.startswith(';') and not l.endswith(':') and not l.startswith('.'):
                body.append(l.split()[0])
            j += 1
        c = Counter(body)
        tot = lambda p: sum(v for k, v in c.items() if re.search(p, k))
        short = subprocess.check_output(['c++filt', name]).decode().split('(')[0]
        print(f'{short[:50]:50s} instr {len(body):6d} mfma {tot("mfma"):4d} accvgpr {tot("accvgpr"):4d} scratch {tot("scratch"):3d} '
              f'waitcnt {c.get("s_waitcnt", 0):4d} ds_r {tot("^ds_read"):4d} ds_w {tot("^ds_write"):4d} gload {tot("^global_load"):4d} '
              f'gstore {tot("^global_store"):4d} branch {tot("^s_cbranch"):4d} bperm {tot("bpermute"):3d}')
        i = j
    i += 1
