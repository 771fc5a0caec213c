#!/bin/bash
# world size 2 on the ONE GPU of a box (both ranks on cuda:0, collective through host memory): functional run of every world > 1 branch
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_world2; mkdir -p $O
timeout 600 python bench.py --gpus 2 --dist-backend gloo --global-batch 3 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/world2_uneven.json 2> $O/world2_uneven.err
timeout 600 python bench.py --gpus 2 --dist-backend gloo --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/world2_even.json 2> $O/world2_even.err
timeout 600 python bench.py --gpus 1 --self-launch --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/world1_rccl.json 2> $O/world1_rccl.err
for f in $O/*.json; do python -c "
import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); c=d['config']; print('$f', '%.1f maps/s'%d['value'], c['world_size'], c['dist_backend'], c['ranks_per_device'], c['global_batch'], c['collective'], c['numa_node_bound'], d['power']['power_w'])"; done
tail -3 $O/world2_uneven.err
