"""Mints tests/golden/ref_stage_vectors.npz from the REFERENCE's own shaders compiled for the CPU (oracle/_ref/libref_stages.so,
built from /root/reference by oracle/kat/build_ref_stages.sh) — run in the authoring container:

    python tests/golden/make_ref_stage_vectors.py

Per golden case (tests/test_ref_stages.py GOLDEN_CASES) and frame, every screen-space buffer as 32-bit words: G-buffer, motion
vectors, direct + indirect reservoirs, both result images, the four filter temporaries.  The oracle (CPU, anywhere) and the HIP
path (GPU box) are held to this file bit for bit.  The file holds numbers only; no reference source is stored.
"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import test_ref_stages as T  # noqa: E402
from helpers import abi  # noqa: E402
from oracle import ref_binding  # noqa: E402


def main():
    assert ref_binding.build(), "needs /root/reference (authoring container)"
    out = {}
    for name in sorted(T.GOLDEN_CASES):
        sc, env, st, W, H, frames, moving = T.golden_setup(name)
        r = ref_binding.Reference(); r.upload_scene(sc.desc(env)); r.resize(W, H)
        def grab(f):
            for buf in T.all_buffers(f):
                out[f"{name}/f{f}/{abi.BUFFER_NAMES[buf]}"] = r.readback(buf).view(np.uint32).copy()
        T.drive([r], sc, st, W, H, frames, moving, None, grab)
    np.savez_compressed(T.VECTORS, **out)
    print("wrote", T.VECTORS, os.path.getsize(T.VECTORS), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
