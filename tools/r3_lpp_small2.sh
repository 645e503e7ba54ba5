#!/bin/bash
# where do 16 lanes per particle overtake 32 on the compact kernels?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_lppsmall; mkdir -p $O
run() { python bench.py --workload $1 --lpp $4 --steps $2 --warmup $3 --no-cpu-baseline --no-aux --profile-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); w=d['config']['workload']; nf=w.split('n_fluid=')[1].split(',')[0]; print('$1', 'n_fluid', nf, 'lpp', d['config']['lanes_per_particle'], f\"{1e3*d['ms_per_step']:.2f} us/step\")"; }
for rep in 1 2 3; do
for wl in "dp=0.04,DL=3" "dp=0.035,DL=3" "dp=0.03,DL=3" "dp=0.0275,DL=3" "dp=0.025,DL=3" "dp=0.0225,DL=3"; do
  for l in 32 16; do run $wl 4000 400 $l; done
done; done 2>&1 | tee $O/lpp2.txt
for l in 32 16; do python bench.py --lpp $l --steps 20 --warmup 5 --no-cpu-baseline --no-aux --profile-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('C2 20/5 lpp', d['config']['lanes_per_particle'], f\"{1e3*d['ms_per_step']:.2f} us/step\")"; done | tee -a $O/lpp2.txt
