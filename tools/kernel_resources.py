"""Compiler-reported resources of every kernel of deeptables_amd/csrc (no GPU needed): VGPRs, AGPRs, spills, scratch, static
LDS and the occupancy the register count allows, from `hipcc --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage`.
python tools/kernel_resources.py [file.hip ...] > profiles/rNN_kernel_resources.txt
(dynamic LDS is set at launch: the `lds` column is the static part only; see the kernels' host code for the rest)"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def demangle(names):
    try:
        out = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True,
                             check=True).stdout.split('\n')
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def resources(path):
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-I', os.path.join(ROOT, 'include'), '-I',
               os.path.join(ROOT, 'deeptables_amd', 'csrc'), '-c', path, '-o', os.path.join(tmp, 'o.o'),
               '-Rpass-analysis=kernel-resource-usage']
        err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = {}, None
    for line in err.split('\n'):
        m = re.search(r'Function Name: (\S+)', line)
        if m:
            cur = rows.setdefault(m.group(1), {})
            continue
        m = re.search(r'remark:\s+([A-Za-z \[\]/]+): (\d+)', line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return rows


def main():
    import __graft_entry__ as ge
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, 'deeptables_amd', 'csrc', '*.hip')))
    print(f'# kernel resources, gfx950, hipcc -O3; source hash {ge.source_hash()}')
    print(f'# {"file":14s} {"vgpr":>5s} {"agpr":>5s} {"spill":>5s} {"scratch":>7s} {"lds":>6s} {"occ":>3s}  kernel')
    for f in files:
        rows = resources(f)
        names = demangle(list(rows))
        for mangled, r in sorted(rows.items(), key=lambda kv: names[kv[0]]):
            name = re.sub(r'\(.*', '', names[mangled]).replace('void ', '')
            print(f'  {os.path.basename(f):14s} {r.get("VGPRs", 0):5d} {r.get("AGPRs", 0):5d} {r.get("VGPRs Spill", 0):5d} '
                  f'{r.get("ScratchSize [bytes/lane]", 0):7d} {r.get("LDS Size [bytes/block]", 0):6d} '
                  f'{r.get("Occupancy [waves/SIMD]", 0):3d}  {name}')


if __name__ == '__main__':
    main()
