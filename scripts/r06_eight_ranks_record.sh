#!/bin/bash
# round 6 (end): the driver's `--gpus 8` command line with all ranks on ONE device over gloo, after the thread counts moved to rt_cpu_budget() — wall time of the whole command
R=$GRAFT_REPO_ROOT; T=${1:-r06_eight}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
s=$(date +%s)
HSA_ENABLE_IPC_MODE_LEGACY=0 RESTIR_BENCH_SHARE_DEVICE=1 RESTIR_DIST_BACKEND=gloo timeout 1700 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 8 --steps 10 --warmup 3 > $O/bench_eight_ranks_shared_device.json 2> $O/eight.err
echo "exit $? wall $(( $(date +%s) - s )) s"
python -c "
import json; d=json.loads([l for l in open('$O/bench_eight_ranks_shared_device.json') if l.startswith('{')][-1]); print({k: d.get(k) for k in ('value','ms_per_step','host','faster_host','wall_s','tiled_equals_untiled','hosts_all_verified')}); print({k: (v if not isinstance(v, dict) else {q: v[q] for q in list(v)[:8]}) for k, v in (d.get('hosts') or {}).items()})"
tail -5 $O/eight.err | cut -c1-300
