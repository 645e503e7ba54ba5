#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -3
timeout -k 10 120 python3 bench.py --steps 20 --warmup 5 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], sorted(d['roofline'].keys()))"
