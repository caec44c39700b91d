#!/bin/bash
# round 4, call 22: where bench.py's layer-by-layer models fault (python -X faulthandler) and which change it follows
# (DT_AMD_CAPTURE_GC=1 left the cyclic collector ON during captures in the tree of that moment — a bisect switch that is gone:
#  the captures now freeze the old objects instead, deeptables_amd/compiled.py)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c22
O=gpurun_out/r4c22
DT_AMD_CAPTURE_GC=1 DT_AMD_ELECT_ROWS=0 python -X faulthandler bench.py --model AutoInt --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/a.json 2> $O/a.err; echo "gc on, elect off: rc $?"
DT_AMD_CAPTURE_GC=1 python -X faulthandler bench.py --model AutoInt --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/b.json 2> $O/b.err; echo "gc on, elect on: rc $?"
DT_AMD_ELECT_ROWS=0 python -X faulthandler bench.py --model AutoInt --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/c.json 2> $O/c.err; echo "gc off, elect off: rc $?"
DT_AMD_ELECT_ROWS=0 python -X faulthandler bench.py --model AutoInt --steps 20 --warmup 5 --steps-per-graph 1 --no-cpu-baseline --no-parity > $O/d.json 2> $O/d.err; echo "gc off, elect off, k=1: rc $?"
DT_AMD_ELECT_ROWS=0 python -X faulthandler bench.py --model AutoInt --steps 20 --warmup 5 --no-cpu-baseline --no-parity --batch 4096 > $O/e.json 2> $O/e.err; echo "gc off, elect off, B=4096: rc $?"
