"""round 6: where the CPU oracle's wall time goes at 1 / 32 / 128 / all threads — per dispatch, wall and process CPU seconds (all threads), rows 272..784 of the headline frame.
   python scripts/r06_oracle_scaling.py [--scale S]      (CPU only; the numbers of interest are the GPU box's 256 host threads)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import restir_amd  # noqa: F401
from restir_amd import abi, host
from oracle.binding import Oracle

scale = float(sys.argv[sys.argv.index("--scale") + 1]) if "--scale" in sys.argv else 1.0
W, H = 1920, 1080
scene = host.Scene().makeProcedural(abi.PROC_BISTRO_EXT_REAL, scale, 1)
env = host.HdrSampling(); env.makeSyntheticSky(2048, 1024, 5e4, 7)
st = host.default_state(W, H, scene, env)
scene.updateCamera(W, H); scene.updateCamera(W, H)
desc = scene.desc(env)
o = Oracle(0); o.upload_scene(desc)
ncpu = len(os.sched_getaffinity(0))
st.time = 1020
Y0, Y1 = 272, 784
STAGES = [("direct", abi.STAGE_DIRECT, [0], False), ("indirect", abi.STAGE_INDIRECT, [0], True), ("denoise_direct x4", abi.STAGE_DENOISE_DIRECT, [0, 1, 2, 3], False),
          ("denoise_indirect x5", abi.STAGE_DENOISE_INDIRECT, [0, 1, 2, 3, 4], True), ("compose", abi.STAGE_COMPOSE, [0], False)]
for threads in sorted({1, 8, min(32, ncpu), min(128, ncpu), ncpu}):
    for pin in ((True, False) if threads > 1 else (True,)):
        y0, y1 = (Y0, Y1) if threads > 1 else (524, 532)
        o.set_threads(threads, pin=pin)
        best = None
        for rep in range(2):
            o.resize(W, H); o.set_camera(scene.getCamera()); o.reset_counters()
            rows = []
            for name, stage, levels, half in STAGES:
                w0, c0 = time.perf_counter(), time.process_time()
                for l in levels:
                    o.run_stage(st, 20, stage, l, y0 // 2 if half else y0, y1 // 2 if half else y1)
                rows.append((name, time.perf_counter() - w0, time.process_time() - c0))
            c = o.counters(); rays = c.closestHitRays + c.anyHitRays
            wall = sum(r[1] for r in rows)
            if best is None or wall < best[0]: best = (wall, rows, rays)
        wall, rows, rays = best
        print(f"threads {threads:3d} pinned {pin!s:5}  rows {y0}..{y1}  {rays} rays  wall {wall:.3f} s  {rays / wall / 1e6:.3f} Mrays/s  | " +
              "  ".join(f"{n}: {w:.3f} s wall, {cpu:.2f} CPU s ({cpu / max(w, 1e-9):.0f} busy)" for n, w, cpu in rows), flush=True)
