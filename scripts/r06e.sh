#!/bin/bash
# r06e: changed GPU tests; builder on the box; persistent waves under PMC; typed vs generic accessors in the wave profile; emulation set with the round-6 stream layout;
# the glTF path at BASELINE scale; the driver's 8-rank command line on one device with its wall time
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06e; mkdir -p $O; cd $R
(timeout 1500 python -m pytest tests/test_gpu_bench_cli.py tests/test_gpu_tiled_rccl.py tests/test_gpu_stream_prio.py tests/test_gpu_mgpu.py "tests/test_gpu_fullsize_allstages.py::test_config3_sponza_1k_textures_1080p_all_stages" -m gpu -x -q > $O/gputests.log 2>&1; grep -n "passed\|failed\|Error" $O/gputests.log | tail -5)
echo "== builder"; bash scripts/r06_bvh_build.sh r06e_bvh 2>&1 | head -24
echo "== persistent waves under PMC"; bash scripts/r06_persist_pmc.sh r06e_persist_pmc
echo "== via glTF at BASELINE scale"
timeout 1500 python bench.py --via-gltf --no-cpu-baseline > $O/bench_via_gltf.json 2> $O/bench_via_gltf.err || tail -5 $O/bench_via_gltf.err
python -c "
import json; d=json.loads(open('$O/bench_via_gltf.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], json.dumps(d['via_gltf'])[:900])"
echo "== emulation set (default levels)"
for n in 2 4; do timeout 900 python bench.py --emulate-world $n --steps 30 --warmup 12 --no-cpu-baseline > $O/emulate${n}_1080p.json 2> $O/e.err; done
timeout 1200 python bench.py --emulate-world 8 --steps 30 --warmup 12 --solo-fresh 2 --no-cpu-baseline > $O/emulate8_1080p.json 2> $O/e.err
for n in 4 8; do timeout 1800 python bench.py --emulate-world $n --width 3840 --height 2160 --steps 20 --warmup 10 --no-cpu-baseline > $O/emulate${n}_4k.json 2> $O/e.err; done
for f in $O/emulate*.json; do python - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], "one", d["single_gpu_serial_ms"], d["single_gpu_frames_in_flight_ms"], "serial slowest", d["slowest_rank_ms"], "period", d.get("rank_period_ms"), "x", d.get("projected_speedup_period_vs_single_gpu_frames_in_flight"), "steady MB", [round(x/1e6,1) for x in d.get("rank_bytes_steady",[])], "xgmi", d.get("xgmi_model",{}).get("on_frame_path_ms"))
for s in d.get("solo_fresh", []): print("   fresh process: rank", s["rank"], "in process", s["period_in_process_ms"], "fresh", s.get("period_fresh_process_ms"), s.get("passes_ms"))
PY
done
echo "== the driver's command line, 8 ranks on this one device (gloo), wall time"
s=$(date +%s)
HSA_ENABLE_IPC_MODE_LEGACY=0 RESTIR_BENCH_SHARE_DEVICE=1 RESTIR_DIST_BACKEND=gloo timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 8 --steps 10 --warmup 3 > $O/bench_eight_ranks_shared_device.json 2> $O/eight.err
echo "exit $? wall $(( $(date +%s) - s )) s"
python -c "
import json; d=json.loads([l for l in open('$O/bench_eight_ranks_shared_device.json') if l.startswith('{')][-1]); print({k: d.get(k) for k in ('value','ms_per_step','host','faster_host','wall_s','tiled_equals_untiled','hosts_all_verified','stream_layout')}); print(d.get('hosts'))"
echo "== typed vs generic accessors in the wave profile"; bash scripts/r06_addrspace_wave.sh r06e_as > /dev/null 2>&1; cat $R/gpurun_out/r06e_as/summary.txt | head -80
