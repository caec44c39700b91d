#!/bin/bash
# kernel resource usage of one csrc/*.hip file (VGPRs / AGPRs / spills / scratch / occupancy per kernel), e.g.
#   tools/kres.sh deepfm 'k_mlp_fwd3<7|k_wgrad'
f=$1; pat=${2:-.}
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DDT_UNUSED -x hip -c /root/repo/deeptables_amd/csrc/$f.hip -o /tmp/kres_$f.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re,subprocess
cur=None;rows=[]
for l in sys.stdin:
    if ' error' in l or 'warning:' in l: print(l.rstrip())
    m=re.search(r'remark:\s+(?:\S+:\d+:\d+:\s+)?(.*?): (\S+) \[-Rpass', l)
    if not m: continue
    k,v=m.group(1).strip(),m.group(2)
    if k.endswith('Name'):
        cur={'name':v}; rows.append(cur)
    elif cur is not None: cur[k]=v
for r in rows:
    try: n=subprocess.check_output(['c++filt',r['name']]).decode().strip()
    except Exception: n=r['name']
    n=re.sub(r'\(.*','',n)
    if re.search(r'$pat', n): print(f\"{n:56s} VGPR {r.get('VGPRs','?'):>4} AGPR {r.get('AGPRs','?'):>4} spill {r.get('VGPRs Spill','?'):>3} scratch {r.get('ScratchSize [bytes/lane]','?'):>4} sgpr {r.get('TotalSGPRs','?'):>4} occ {r.get('Occupancy [waves/SIMD]','?')}\")
"
