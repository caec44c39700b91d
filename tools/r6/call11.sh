#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6c11; mkdir -p $O
sed -i 's/const int join_env = 1; /const int join_env = 0; /' deeptables_amd/csrc/deepfm.hip
grep -n "const int join_env" deeptables_amd/csrc/deepfm.hip
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1; tail -1 $O/build.txt
timeout 900 python tools/r6/dbg_eager_repeat.py > $O/dbg_nojoin.txt 2>&1
tail -5 $O/dbg_nojoin.txt
