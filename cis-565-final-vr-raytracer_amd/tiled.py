"""Row-tiled multi-GPU frame: one process per GPU (torch.distributed; backend "nccl" == RCCL over xGMI).

The reference is single-GPU; BASELINE.json asks for the frame to be row-tiled over the GPUs of one node.
Pixels are independent inside a stage except for:
  * temporal reprojection  -> reads last frame's G-buffer / reservoirs at an arbitrary pixel
  * A-Trous taps           -> +-2*2^level rows of the level's input image and of the G-buffer
  * compose / indirect     -> coord/2 <-> 2*coord (stay inside a band whose height is a multiple of 16)

Rank r owns the full-resolution rows [r*B, min((r+1)*B, H)), B = 16*ceil(ceil(H/16)/world), and the half-resolution rows
[r*B/2, ...).  Every rank keeps full-size buffers (scene, BVH8, textures and screen-space state are replicated: the
frame state is ~0.5 GB at 1080p against 288 GB of HBM); RNG seeds use global pixel indices, so the tiled frame is
bit-identical to the untiled one.  Per frame:

   direct stage (band)
   all-gather this G-buffer ........................ async, overlaps the indirect stage; needed by the denoiser halos
   all-gather direct reservoirs + light ids ........ async, consumed by NEXT frame's temporal reuse
   indirect stage (half band)
   all-gather indirect reservoirs .................. async, consumed by NEXT frame
   wait(G-buffer)
   4 x [halo exchange of the level's input (2*2^l rows per neighbour) ; denoise-direct level l (band)]
   5 x [halo exchange (half-res) ; denoise-indirect level l (half band)]
   compose (band)
   gather the two result images to rank 0 .......... async
   (next frame starts by waiting for the history all-gathers)

All-gathers are in place on the ctx-owned HBM buffers (rt_device_ptr; the allocations carry slack rows so world*B rows
fit).  xGMI is point-to-point: halo traffic uses only the two neighbour links, all-gathers use all of them.
"""
import math

from . import abi

_COLOR_BYTES = 16


def band_height(H, world):
    units = (H + 15) // 16
    return 16 * int(math.ceil(units / float(world)))


def band_rows(H, world, rank):
    B = band_height(H, world)
    return min(rank * B, H), min((rank + 1) * B, H)


class LocalComm:
    """world == 1: every exchange is a no-op."""
    rank, world = 0, 1
    def all_gather_rows(self, tensor, chunk_bytes, async_op=False): return None
    def halo_exchange(self, tensor, pitch, y0, y1, halo, H, B): return None
    def gather_rows_to(self, tensor, chunk_bytes, dst=0, async_op=False): return None
    def wait(self, work): pass
    def barrier(self): pass


class TorchComm:
    """torch.distributed collectives on flat uint8 tensors that alias the renderer's buffers."""
    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.nccl = dist.get_backend(group) == "nccl"

    def all_gather_rows(self, tensor, chunk_bytes, async_op=False):
        out = tensor[: self.world * chunk_bytes]
        mine = out[self.rank * chunk_bytes:(self.rank + 1) * chunk_bytes]
        if self.nccl:  # in place: input is the rank-th chunk of the output
            return self.dist.all_gather_into_tensor(out, mine, group=self.group, async_op=async_op)
        chunks = [out[i * chunk_bytes:(i + 1) * chunk_bytes] for i in range(self.world)]
        return self.dist.all_gather(chunks, mine.clone(), group=self.group, async_op=async_op)

    def halo_exchange(self, tensor, pitch, y0, y1, halo, H, B):
        """Fill rows [y0-halo, y0) and [y1, y1+halo) from the neighbouring bands.  Falls back to an all-gather when
        a halo is taller than a neighbour's band (tiny images / many ranks)."""
        if y1 <= y0:
            # a rank without rows still has to take part in collective fallbacks; p2p needs nothing from it
            pass
        if halo > B or (H - (self.world - 1) * B) < min(halo, H):  # last band may be short
            w = self.all_gather_rows(tensor, B * pitch, async_op=False)
            return w
        ops = []
        P2P = self.dist.P2POp
        up, down = self.rank - 1, self.rank + 1
        if y1 > y0:
            if up >= 0 and y0 > 0:
                n = min(halo, y1 - y0)
                ops.append(P2P(self.dist.isend, tensor[y0 * pitch:(y0 + n) * pitch], up, group=self.group))
                ops.append(P2P(self.dist.irecv, tensor[(y0 - halo) * pitch:y0 * pitch], up, group=self.group))
            if down < self.world and y1 < H:
                n = min(halo, H - y1)
                ops.append(P2P(self.dist.isend, tensor[(y1 - min(halo, y1 - y0)) * pitch:y1 * pitch], down, group=self.group))
                ops.append(P2P(self.dist.irecv, tensor[y1 * pitch:(y1 + n) * pitch], down, group=self.group))
        if not ops:
            return None
        works = self.dist.batch_isend_irecv(ops)
        for w in works:
            w.wait()
        return None

    def gather_rows_to(self, tensor, chunk_bytes, dst=0, async_op=False):
        # NCCL has no in-place gather primitive; an all-gather keeps one code path and every rank ends with the frame
        return self.all_gather_rows(tensor, chunk_bytes, async_op=async_op)

    def wait(self, work):
        if work is not None:
            work.wait()

    def barrier(self):
        self.dist.barrier(group=self.group)


class TiledFrame:
    """Drives one backend (HIP Renderer or, in the CPU tests, the oracle) over this rank's row band.

    backend needs: run_stage(state, frames, stage, level, row_begin, row_end) and tensor(buf) -> (flat uint8 torch
    tensor over the whole allocation, row pitch in bytes)."""

    def __init__(self, backend, comm, width, height):
        self.b, self.comm, self.W, self.H = backend, comm, width, height
        self.B = band_height(height, comm.world)
        if comm.world > 8 and (comm.world * self.B - height) > 128:
            raise ValueError("buffer slack rows (128) do not cover this world size")
        self.y0, self.y1 = band_rows(height, comm.world, comm.rank)
        self.Hh, self.Bh = height // 2, self.B // 2
        self.h0, self.h1 = min(comm.rank * self.Bh, self.Hh), min((comm.rank + 1) * self.Bh, self.Hh)
        self._pending = []  # history gathers of the previous frame

    def _t(self, buf):
        return self.b.tensor(buf)

    def render_frame(self, state, frames):
        c, b = self.comm, self.b
        cur = frames & 1
        for w in self._pending:  # last frame's reservoirs / light ids must have landed before temporal reuse reads them
            c.wait(w)
        self._pending = []

        run = lambda stage, level, r0, r1: (b.run_stage(state, frames, stage, level, r0, r1) if r1 > r0 else None)  # noqa: E731
        run(abi.STAGE_DIRECT, 0, self.y0, self.y1)
        g, gp = self._t(abi.BUF_GBUFFER0 + cur)
        wG = c.all_gather_rows(g, self.B * gp, async_op=True)
        for buf in (abi.BUF_DIRECT_RESV0 + cur, abi.BUF_LIGHT_ID0 + cur):
            t, p = self._t(buf)
            self._pending.append(c.all_gather_rows(t, self.B * p, async_op=True))

        run(abi.STAGE_INDIRECT, 0, self.h0, self.h1)
        t, p = self._t(abi.BUF_INDIRECT_RESV0 + cur)
        self._pending.append(c.all_gather_rows(t, self.Bh * p, async_op=True))

        c.wait(wG)
        if state.denoise > 0:
            pitch = self.W * _COLOR_BYTES
            # DirectResult -> DirA -> DirB -> DirA -> DirectResult (denoise_direct.comp:152-172)
            src = [abi.BUF_DIRECT_RESULT0 + cur, abi.BUF_DENOISE_DIR_A, abi.BUF_DENOISE_DIR_B, abi.BUF_DENOISE_DIR_A]
            for l in range(4):
                t, _ = self._t(src[l])
                c.halo_exchange(t, pitch, self.y0, self.y1, 2 << l, self.H, self.B)
                run(abi.STAGE_DENOISE_DIRECT, l, self.y0, self.y1)
            # IndA -> IndB -> IndA -> thisIndirectResult -> IndA -> IndB (denoise_indirect.comp:146-171); images keep the full-res pitch
            src = [abi.BUF_DENOISE_IND_A, abi.BUF_DENOISE_IND_B, abi.BUF_DENOISE_IND_A, abi.BUF_INDIRECT_RESULT0 + cur, abi.BUF_DENOISE_IND_A]
            for l in range(5):
                t, _ = self._t(src[l])
                c.halo_exchange(t, pitch, self.h0, self.h1, 2 << l, self.Hh, self.Bh)
                run(abi.STAGE_DENOISE_INDIRECT, l, self.h0, self.h1)
        run(abi.STAGE_COMPOSE, 0, self.y0, self.y1)
        works = []
        for buf in (abi.BUF_DIRECT_RESULT0 + cur, abi.BUF_INDIRECT_RESULT0 + cur):
            t, p = self._t(buf)
            works.append(c.gather_rows_to(t, self.B * p, dst=0, async_op=True))
        self._result_pending = works

    def finish(self):
        """Wait for everything in flight (call before reading results / at the end of a timed region)."""
        for w in self._pending + getattr(self, "_result_pending", []):
            self.comm.wait(w)
        self._pending, self._result_pending = [], []


class RendererTensors:
    """Backend adapter: HIP Renderer + torch views of its HBM buffers (zero copy through __cuda_array_interface__)."""
    def __init__(self, renderer):
        import torch
        self.r, self.torch = renderer, torch
        self._cache = {}
    def run_stage(self, state, frames, stage, level, r0, r1):
        self.r.run_stage(state, frames, stage, level, r0, r1)
    def tensor(self, buf):
        if buf not in self._cache:
            arr, pitch = self.r.device_array(buf)
            self._cache[buf] = (self.torch.as_tensor(arr, device=f"cuda:{self.torch.cuda.current_device()}"), pitch)
        return self._cache[buf]
