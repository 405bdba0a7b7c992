"""Row-tiled multi-GPU frame: one process per GPU (torch.distributed; backend "nccl" == RCCL over xGMI).

The reference is single-GPU; BASELINE.json asks for the frame to be row-tiled over the GPUs of one node.
Pixels are independent inside a stage except for:
  * temporal reprojection  -> reads last frame's G-buffer / reservoirs at the reprojected pixel
  * A-Trous taps           -> +-2*2^level rows of the level's input image and of the G-buffer
  * compose / indirect     -> coord/2 <-> 2*coord (stay inside a band whose height is a multiple of 16)

Rank r owns the full-resolution rows [r*B, min((r+1)*B, H)), B = 16*ceil(ceil(H/16)/world), and the half-resolution rows
[r*B/2, ...).  Scene, BVH8, textures and full-size screen-space buffers are replicated (0.5 GB of frame state against 288 GB
of HBM); RNG seeds use global pixel indices, so the tiled frame is bit-identical to the untiled one.

Communication is sized for a frame that takes a few hundred microseconds per GPU — three batched exchanges per frame:

  1. (start of frame) wait for the HISTORY HALO sent at the end of the previous frame: the neighbours' HIST_HALO rows of last
     frame's G-buffer, direct reservoirs, light ids and indirect reservoirs.  Temporal reuse normally reprojects within a
     few rows of the pixel; a lookup that lands outside band+halo raises a flag on the GPU (rt_set_history_rows /
     rt_history_miss).  The flag is max-reduced over ranks; if anybody missed, the full history is all-gathered and the two
     ray-traced stages are run again (rare: fast camera motion) — so the result is exact for any camera path.
  2. after the direct + indirect stages, ONE neighbour exchange carries everything the 9 A-Trous passes need: 136 G-buffer
     rows, 40 rows of noisy direct colour, 72 half-res rows of noisy indirect colour.  Each rank then filters a region that
     starts wider than its band and shrinks by the next levels' reach (direct +32/+24/+16/+0 rows, indirect
     +64/+56/+48/+32/+0): the overlap is recomputed redundantly (+17 % of a cheap stage) instead of exchanging 9 times.
  3. compose (band), then send the history halo for the next frame and the band's two result images to rank 0 (async).

xGMI is point-to-point (7 links x ~153 GB/s per GPU): the halos use only the two neighbour links and are 4-8 MB per
frame per neighbour; nothing is all-gathered in the steady state.  When a halo is taller than a neighbour's band (tiny
images, many ranks) the exchange falls back to an in-place all-gather (allocations carry slack rows for that).
"""
import math

from . import abi

_COLOR_BYTES = 16
HIST_HALO = 32                      # full-res rows of last-frame history kept from each neighbour (multiple of 16)
DIRECT_GROW = (32, 24, 16, 0)       # rows added on each side of the band for A-Trous level l's output (multiples of 8)
INDIRECT_GROW = (64, 56, 48, 32, 0)
HALO_DIRECT_COLOR = 40              # >= DIRECT_GROW[0] + 2
HALO_INDIRECT_COLOR = 72            # half-res rows, >= INDIRECT_GROW[0] + 2
HALO_GBUFFER = 144                  # full-res rows, >= 2 * (INDIRECT_GROW[0] + 2) and a multiple of 16


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
    def halo_exchange(self, items, async_op=False): return []
    def gather_rows_to(self, tensor, chunk_bytes, dst=0, async_op=False): return None
    def any_flag(self, flag): return bool(flag)
    def wait(self, work): pass
    def barrier(self): pass


class TorchComm:
    """torch.distributed collectives on flat uint8 tensors that alias the renderer's buffers."""
    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.nccl = dist.get_backend(group) == "nccl"
        self._flag = torch.zeros(1, dtype=torch.int32, device="cuda" if self.nccl else "cpu")

    def all_gather_rows(self, tensor, chunk_bytes, async_op=False):
        out = tensor[: self.world * chunk_bytes]
        mine = out[self.rank * chunk_bytes:(self.rank + 1) * chunk_bytes]
        if self.nccl:  # in place: input is the rank-th chunk of the output
            return self.dist.all_gather_into_tensor(out, mine, group=self.group, async_op=async_op)
        chunks = [out[i * chunk_bytes:(i + 1) * chunk_bytes] for i in range(self.world)]
        return self.dist.all_gather(chunks, mine.clone(), group=self.group, async_op=async_op)

    def halo_exchange(self, items, async_op=False):
        """items: [(tensor, pitch, y0, y1, halo, H, B)].  For every item fill rows [y0-halo, y0) and [y1, y1+halo) (clipped
        to [0,H)) from the neighbouring bands, all items in ONE batched send/recv.  Items whose halo does not fit in a single
        neighbour band fall back to an all-gather.  Returns the list of pending works."""
        works, ops = [], []
        P2P = self.dist.P2POp
        for (tensor, pitch, y0, y1, halo, H, B) in items:
            if halo <= 0:
                continue
            if halo > B:                                 # a halo would span more than one neighbour band
                works.append(self.all_gather_rows(tensor, B * pitch, async_op=async_op))
                continue
            if y1 <= y0:
                continue
            # every band except the last non-empty one is B >= halo rows tall; only the last may be shorter, and it has no
            # neighbour below, so "send min(halo, own height) rows" always matches what the receiver expects
            ns = min(halo, y1 - y0)
            up, down = self.rank - 1, self.rank + 1
            if up >= 0 and y0 > 0:
                ops.append(P2P(self.dist.isend, tensor[y0 * pitch:(y0 + ns) * pitch], up, group=self.group))
                ops.append(P2P(self.dist.irecv, tensor[(y0 - halo) * pitch:y0 * pitch], up, group=self.group))
            if down < self.world and y1 < H:
                n = min(halo, H - y1)
                ops.append(P2P(self.dist.isend, tensor[(y1 - ns) * pitch:y1 * pitch], down, group=self.group))
                ops.append(P2P(self.dist.irecv, tensor[y1 * pitch:(y1 + n) * pitch], down, group=self.group))
        if ops:
            works += self.dist.batch_isend_irecv(ops)
        works = [w for w in works if w is not None]
        if not async_op:
            for w in works:
                w.wait()
            return []
        return works

    def gather_rows_to(self, tensor, chunk_bytes, dst=0, async_op=False):
        # NCCL has no in-place gather primitive; grouped send/recv to dst: only dst's links carry the traffic
        P2P = self.dist.P2POp
        if self.rank == dst:
            ops = [P2P(self.dist.irecv, tensor[r * chunk_bytes:(r + 1) * chunk_bytes], r, group=self.group) for r in range(self.world) if r != dst]
        else:
            ops = [P2P(self.dist.isend, tensor[self.rank * chunk_bytes:(self.rank + 1) * chunk_bytes], dst, group=self.group)]
        works = self.dist.batch_isend_irecv(ops) if ops else []
        if not async_op:
            for w in works:
                w.wait()
            return None
        return works

    def any_flag(self, flag):
        self._flag.fill_(1 if flag else 0)
        self.dist.all_reduce(self._flag, op=self.dist.ReduceOp.MAX, group=self.group)
        return bool(int(self._flag.item()))

    def wait(self, work):
        if work is None:
            return
        for w in (work if isinstance(work, (list, tuple)) else [work]):
            if w is not None:
                w.wait()

    def barrier(self):
        self.dist.barrier(group=self.group)


class TiledFrame:
    """Drives one backend (HIP Renderer or, in the CPU tests, the oracle) over this rank's row band.

    backend needs: run_stage(state, frames, stage, level, row_begin, row_end), tensor(buf) -> (flat uint8 torch tensor over the
    whole allocation, row pitch in bytes), set_history_rows(r0, r1), history_miss() -> bool (synchronising)."""

    def __init__(self, backend, comm, width, height):
        self.b, self.comm, self.W, self.H = backend, comm, width, height
        self.B = band_height(height, comm.world)
        if comm.world * self.B - height > 128:
            raise ValueError("buffer slack rows (128) do not cover this world size")
        self.y0, self.y1 = band_rows(height, comm.world, comm.rank)
        self.Hh, self.Bh = height // 2, self.B // 2
        self.h0, self.h1 = min(comm.rank * self.Bh, self.Hh), min((comm.rank + 1) * self.Bh, self.Hh)
        self._pending = []          # history halos in flight (sent at the end of the previous frame)
        self._result_pending = []
        self.history_fallbacks = 0  # frames that needed the full history (statistics)

    def _t(self, buf):
        return self.b.tensor(buf)

    def _run(self, state, frames, stage, level, r0, r1, limit):
        r0, r1 = max(0, r0), min(limit, r1)
        if r1 > r0:
            self.b.run_stage(state, frames, stage, level, r0, r1)

    def _traced_stages(self, state, frames):
        self._run(state, frames, abi.STAGE_DIRECT, 0, self.y0, self.y1, self.H)
        self._run(state, frames, abi.STAGE_INDIRECT, 0, self.h0, self.h1, self.Hh)

    def render_frame(self, state, frames):
        c, b = self.comm, self.b
        cur, last = frames & 1, (frames + 1) & 1
        single = c.world == 1
        if not single and state.ReSTIRState in (abi.RESTIR_SPATIAL, abi.RESTIR_SPATIOTEMPORAL):
            # the spatial reuse step reads the cached reservoirs of the rows above / below the band (direct_stage.comp:86-107);
            # that buffer is not part of the halo exchange
            raise NotImplementedError("ReSTIRState eSpatial / eSpatiotemporal is single-GPU only in this build")
        c.wait(self._pending)
        self._pending = []

        # ---- ray-traced stages on the band, with exact fallback when temporal reuse leaves band + history halo ------------
        if not single:
            b.set_history_rows(max(0, self.y0 - HIST_HALO), min(self.H, self.y1 + HIST_HALO))
        self._traced_stages(state, frames)
        if not single and c.any_flag(b.history_miss()):
            self.history_fallbacks += 1
            for buf, chunk in ((abi.BUF_GBUFFER0 + last, self.B), (abi.BUF_DIRECT_RESV0 + last, self.B), (abi.BUF_LIGHT_ID0 + last, self.B),
                               (abi.BUF_INDIRECT_RESV0 + last, self.Bh)):
                t, p = self._t(buf)
                c.all_gather_rows(t, chunk * p)
            b.set_history_rows(0, self.H)
            self._traced_stages(state, frames)
            b.history_miss()  # clear

        # ---- one neighbour exchange feeds all nine A-Trous passes --------------------------------------------------------
        pitch = self.W * _COLOR_BYTES
        if state.denoise > 0 and not single:
            g, gp = self._t(abi.BUF_GBUFFER0 + cur)
            dcol, _ = self._t(abi.BUF_DIRECT_RESULT0 + cur)
            icol, _ = self._t(abi.BUF_DENOISE_IND_A)
            c.halo_exchange([(g, gp, self.y0, self.y1, HALO_GBUFFER, self.H, self.B),
                             (dcol, pitch, self.y0, self.y1, HALO_DIRECT_COLOR, self.H, self.B),
                             (icol, pitch, self.h0, self.h1, HALO_INDIRECT_COLOR, self.Hh, self.Bh)])
        if state.denoise > 0:
            for l in range(4):
                g_ = 0 if single else DIRECT_GROW[l]
                self._run(state, frames, abi.STAGE_DENOISE_DIRECT, l, self.y0 - g_, self.y1 + g_, self.H)
            for l in range(5):
                g_ = 0 if single else INDIRECT_GROW[l]
                self._run(state, frames, abi.STAGE_DENOISE_INDIRECT, l, self.h0 - g_, self.h1 + g_, self.Hh)
        self._run(state, frames, abi.STAGE_COMPOSE, 0, self.y0, self.y1, self.H)

        # ---- for the next frame: history halo; for the display: result bands to rank 0 (both asynchronous) -----------------
        if not single:
            items = []
            for buf in (abi.BUF_GBUFFER0 + cur, abi.BUF_DIRECT_RESV0 + cur, abi.BUF_LIGHT_ID0 + cur):
                t, p = self._t(buf)
                items.append((t, p, self.y0, self.y1, HIST_HALO, self.H, self.B))
            t, p = self._t(abi.BUF_INDIRECT_RESV0 + cur)
            items.append((t, p, self.h0, self.h1, HIST_HALO // 2, self.Hh, self.Bh))
            self._pending = c.halo_exchange(items, async_op=True)
            c.wait(self._result_pending)
            self._result_pending = []
            for buf in (abi.BUF_DIRECT_RESULT0 + cur, abi.BUF_INDIRECT_RESULT0 + cur):
                t, p = self._t(buf)
                w = c.gather_rows_to(t, self.B * p, dst=0, async_op=True)
                if w:
                    self._result_pending += list(w)

    def finish(self):
        """Wait for everything in flight (call before reading results / at the end of a timed region)."""
        self.comm.wait(self._pending)
        self.comm.wait(self._result_pending)
        self._pending, self._result_pending = [], []


class RendererTensors:
    """Backend adapter: HIP Renderer + torch views of its HBM buffers (zero copy through __cuda_array_interface__)."""
    def __init__(self, renderer):
        import torch
        self.r, self.torch = renderer, torch
        self._cache = {}
    def run_stage(self, state, frames, stage, level, r0, r1):
        self.r.run_stage(state, frames, stage, level, r0, r1)
    def set_history_rows(self, r0, r1):
        self.r.set_history_rows(r0, r1)
    def history_miss(self):
        return self.r.history_miss()
    def tensor(self, buf):
        if buf not in self._cache:
            arr, pitch = self.r.device_array(buf)
            self._cache[buf] = (self.torch.as_tensor(arr, device=f"cuda:{self.torch.cuda.current_device()}"), pitch)
        return self._cache[buf]
