#!/bin/bash
# round 6: why is the BVH2 phase of the builder 1.3 s in a fresh process (scripts/r06_bvh_build.sh) and 4.6 s inside bench.py?  The same build (hash pinned) in
# fresh processes that differ in what they did before it.  usage (gpurun): bash scripts/r06_build_in_process.sh <tag>
R=$GRAFT_REPO_ROOT; T=${1:-r06_build_proc}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
cat > $O/one.py <<'PY'
import ctypes as C, os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
what = sys.argv[1]
if what in ("torch", "torch_alloc", "renderer"):
    import torch
import restir_amd
from restir_amd import abi, host
from restir_amd.renderer import HIP_LIB_PATH, Renderer
sc = host.Scene().makeProcedural(abi.PROC_BISTRO_EXT_REAL, 1.0, 1); desc = sc.desc(None)
if what == "torch":
    torch.cuda.init(); x = torch.zeros(1, device="cuda"); torch.cuda.synchronize()
if what == "torch_alloc":
    torch.cuda.init(); x = torch.zeros(1 << 30, dtype=torch.uint8, device="cuda"); y = torch.zeros(1 << 30, dtype=torch.uint8).pin_memory(); torch.cuda.synchronize()
os.environ["RESTIR_BVH_TIMING"] = "1"
if what == "renderer":
    r = Renderer().setup(0); t0 = time.time(); r.load_scene(desc); print("renderer.load_scene %.3f s" % (time.time() - t0), flush=True)
else:
    L = C.CDLL(HIP_LIB_PATH)
    L.rt_bvh8_build_hash.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    out, sec = (C.c_uint64 * 8)(), C.c_double()
    for rep in range(2):
        assert L.rt_bvh8_build_hash(C.byref(desc), os.cpu_count(), out, C.byref(sec)) == 0
        print("%s: build %.3f s hash %016x" % (what, sec.value, out[0]), flush=True)
PY
for v in fresh torch torch_alloc renderer; do
  echo "== $v"; python $O/one.py $v 2>&1 | grep "build\|load\|hash" | cut -c1-120
done
echo "== renderer, GLIBC_TUNABLES mmap_threshold 1 GiB + trim_threshold 4 GiB"
GLIBC_TUNABLES=glibc.malloc.mmap_threshold=1073741824:glibc.malloc.trim_threshold=4294967296 python $O/one.py renderer 2>&1 | grep "build\|load\|hash" | cut -c1-120
echo "== torch, same tunables"
GLIBC_TUNABLES=glibc.malloc.mmap_threshold=1073741824:glibc.malloc.trim_threshold=4294967296 python $O/one.py torch 2>&1 | grep "build\|load\|hash" | cut -c1-120
echo "== renderer, HSA_XNACK=0"
HSA_XNACK=0 python $O/one.py renderer 2>&1 | grep "build\|load\|hash" | cut -c1-120
