#!/bin/bash
# Round 6, verdict item 4: the BVH8 builder on the GPU box's host cores — phase times (RESTIR_BVH_TIMING), the tree hash for 1 / 16 / 256 builder threads and three
# repeated runs (rt_bvh8_build_hash: must agree), then the frame on the new tree against the library round 5 shipped.  usage (gpurun): bash scripts/r06_bvh_build.sh <tag>
R=$GRAFT_REPO_ROOT; T=${1:-r06bvh}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
python - > $O/build.txt 2> $O/build.err <<'PY'
import ctypes as C, os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import restir_amd
from restir_amd import abi, host
from restir_amd.renderer import HIP_LIB_PATH
L = C.CDLL(HIP_LIB_PATH)
L.rt_bvh8_build_hash.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
sc = host.Scene().makeProcedural(abi.PROC_BISTRO_EXT_REAL, 1.0, 1); desc = sc.desc(None)
print("host threads", os.cpu_count(), "triangles", sc.getStat()["instancedTriangles"], flush=True)
def run(threads, **env):
    os.environ.update(env)
    out, sec = (C.c_uint64 * 8)(), C.c_double()
    assert L.rt_bvh8_build_hash(C.byref(desc), threads, out, C.byref(sec)) == 0
    for k in env: os.environ.pop(k)
    return tuple(out[:6]), sec.value
ref = None
for threads in (os.cpu_count(), os.cpu_count(), os.cpu_count(), 16, 1):
    os.environ["RESTIR_BVH_TIMING"] = "1" if threads == os.cpu_count() else "0"
    h, s = run(threads)
    print("threads %3d  build %.3f s  hash %016x nodes %d leaf records %d splits %d rotations %d depth %d" % ((threads, s) + h), flush=True)
    ref = ref or h
    assert h == ref, "the tree depends on the thread count / the run"
print("identical for every thread count and run: True")
h, s = run(os.cpu_count(), RESTIR_BVH_SPLIT_BUDGET="0.05")
print("binding cap 0.05: build %.3f s (two attempts)  nodes %d leaf records %d splits %d" % (s, h[1], h[2], h[3]))
h, s = run(os.cpu_count(), RESTIR_BVH_SPLIT="0", RESTIR_BVH_ROTATE="0")
print("object splits only, no rotations: build %.3f s  nodes %d leaf records %d" % (s, h[1], h[2]))
PY
cat $O/build.txt; grep "bvh8 build" $O/build.err | head -12
echo "== frame: round-5 library (racy builder) against the tree in the tree"
REPS=2 bash scripts/ab_libs2.sh $T "r05|RESTIR_HIP_LIB=\$R/cis-565-final-vr-raytracer_amd/csrc/_prev/librestir_hip_r05.so" "r06|-"
grep -h bvh8_build_s $O/bench_r05_1.json $O/bench_r06_1.json | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('bvh8_build_s (rt_upload_scene + rt_build_accel)', d['config'].get('bvh8_build_s'), d['config'].get('accel'))"
