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
import os

from . import abi

_COLOR_BYTES = 16
# Full-res rows of last-frame history kept from each neighbour.  Adaptive, the rule of csrc/mgpu.cpp (HIST_HALO_MIN / MAX / CALM): HIST_HALO_MIN rows while no
# temporal lookup leaves band + halo, doubled (up to HIST_HALO_MAX) for the frames after one did, halved again after HIST_HALO_CALM frames without; a lookup
# outside is always caught (exact fallback), so the width only trades bytes against fallbacks.  HIST_HALO = <n> forces a fixed width (tests: 0).
HIST_HALO = None
HIST_HALO_MIN, HIST_HALO_MAX, HIST_HALO_CALM = 16, 64, 16
DIRECT_GROW = (32, 24, 16, 0)       # rows added on each side of the band for A-Trous level l's output (multiples of 8)
INDIRECT_GROW = (64, 56, 48, 32, 0)
HALO_DIRECT_COLOR = 40              # >= DIRECT_GROW[0] + 2
HALO_INDIRECT_COLOR = 72            # half-res rows, >= INDIRECT_GROW[0] + 2
# G-buffer rows the nine filter passes read beyond the band: the direct chain reaches DIRECT_GROW[0] + 2 rows (every row), the indirect chain
# INDIRECT_GROW[0] + 2 half-res rows = 132 full-res rows, of which it reads the EVEN ones only (loadThisGeometry(2q), denoise_common.glsl:42-55)
HALO_GBUFFER_FULL = 40
HALO_GBUFFER = 144                  # full-res rows, >= 2 * (INDIRECT_GROW[0] + 2) and a multiple of 16
HALO_KINDS = ("history", "filter", "spatial", "moved", "gather", "fallback")   # byte accounting by purpose (same names as rt_mgpu_stats::haloBytesKind)


class Halo:
    """One item of a halo exchange: fill `halo` rows of `tensor` (row pitch `pitch`, rows partitioned by `part`, clipped to [0, limit)) above and below this
    rank's band from the ranks that own them.  inner: rows closer than this to the band are NOT part of the item (an outer ring); even: even rows only;
    width: only the first `width` bytes of a row (the half-resolution temporaries live in full-size allocations).  Strided / partial items travel through a
    packed staging tensor (a send / recv needs contiguous memory)."""
    __slots__ = ("tensor", "pitch", "part", "halo", "limit", "inner", "even", "width", "kind")

    def __init__(self, tensor, pitch, part, halo, limit, inner=0, even=False, width=None, kind=None):
        self.tensor, self.pitch, self.part, self.halo, self.limit = tensor, pitch, part, halo, limit
        self.inner, self.even, self.width = inner, even, (None if width is None or width >= pitch else int(width))
        self.kind = kind   # accounting purpose of this item (HALO_KINDS); None: the exchange's

    def need(self, r):
        """row segments rank r needs: [(lo, hi)] above and below its band"""
        y0, y1 = self.part[r], self.part[r + 1]
        if y1 <= y0:
            return []
        return [(max(0, y0 - self.halo), max(0, y0 - self.inner)), (min(self.limit, y1 + self.inner), min(self.limit, y1 + self.halo))]

    def rows(self, a, b):
        """(first row, row count, row step) of the rows of [a, b) this item moves"""
        if self.even:
            a = (a + 1) & ~1
            return a, max(0, (b - a + 1) // 2), 2
        return a, max(0, b - a), 1

    def nbytes(self, a, b):
        _, n, _ = self.rows(a, b)
        return n * (self.width if self.width is not None else self.pitch)


def halo_view(item, a, b, tensor=None):
    """the rows of [a, b) the item moves, as a 2-D (rows, bytes) view of its tensor (or of `tensor`: another rank's copy of the same buffer)"""
    a, n, step = item.rows(a, b)
    if n <= 0:
        return None
    t = item.tensor if tensor is None else tensor
    v = t[a * item.pitch:(a + (n - 1) * step + 1) * item.pitch].view(-1, item.pitch)[::step]
    return v if item.width is None else v[:, :item.width]


class Pending:
    """Works of one batched exchange plus the unpacking of its staged (strided / partial-width) receives.  wait() may be called once per consuming stream:
    the first call waits for the transfers and unpacks on the current stream, later calls (other streams) wait for that unpacking."""
    def __init__(self, comm, works, unpack, keep=(), corrupt=()):
        self.comm, self.works, self.unpack, self.keep, self.event, self.done = comm, works, unpack, list(keep), None, False
        self.corrupt = list(corrupt)   # test hook (RESTIR_TEST_CORRUPT_HALO): received views to damage once they have arrived

    def wait(self):
        if self.done:
            if self.event is not None:
                self.comm.torch.cuda.current_stream().wait_event(self.event)
            return
        self.comm._wait_works(self.works)
        if self.works and not self.comm.nccl and self.comm.torch.cuda.is_available() and self.comm.torch.cuda.is_initialized():
            self.comm.torch.cuda.synchronize()
        for dst, src in self.unpack:
            if self.comm.nccl:
                src.record_stream(self.comm.torch.cuda.current_stream())   # staged on the posting stream, read here on the consumer's: the allocator must not
                                                                           # hand the block out again before this copy has run
            dst.copy_(src)
        for v in self.corrupt:
            v[:1, :64] = 0x3f
        self.corrupt = []
        if self.comm.nccl and (self.works or self.unpack):   # a later wait on another stream orders that stream after transfer + unpacking
            self.event = self.comm.torch.cuda.Event(); self.event.record(self.comm.torch.cuda.current_stream())
        self.done, self.works, self.unpack, self.keep = True, [], [], []


def band_height(H, world):
    units = (H + 15) // 16
    return 16 * int(math.ceil(units / float(world)))


def band_rows(H, world, rank):
    B = band_height(H, world)
    return min(rank * B, H), min((rank + 1) * B, H)


def equal_partition(H, world):
    """row boundaries [b_0 = 0, ..., b_world = H] of the equal-height partition (multiples of 16)"""
    return [min(r * band_height(H, world), H) for r in range(world)] + [H]


def half_partition(part, H):
    """the same partition on the half-resolution grid"""
    Hh = H // 2
    return [min(p // 2, Hh) for p in part[:-1]] + [Hh]


def plan_bands(H, world, stripe_cost, prev=None, max_move=None):
    """Cost-weighted band boundaries: equalise the summed per-stripe cost (a stripe = 16 rows), every rank keeps >= 1 stripe; with
    `prev` a boundary moves at most max_move stripes.  Same rule as csrc/mgpu.cpp rebalance()."""
    stripes = (H + 15) // 16
    cost = [max(1e-9, float(c)) for c in stripe_cost]
    assert len(cost) == stripes and stripes >= world
    total = sum(cost)
    nb, acc, r = [0] * (world + 1), 0.0, 1
    for s_ in range(stripes):
        acc += cost[s_]
        while r < world and acc >= total * r / world:
            nb[r] = s_ + 1; r += 1
    for k in range(r, world):
        nb[k] = stripes
    nb[world] = stripes
    for k in range(1, world):
        v = nb[k]
        if prev is not None and max_move is not None:
            old = prev[k] // 16
            v = max(old - max_move, min(old + max_move, v))
        v = max(v, nb[k - 1] + 1)
        v = min(v, stripes - (world - k))
        nb[k] = v
    return [min(H, b * 16) for b in nb[:-1]] + [H]


def diffuse_bands(part, rank_ms, tol=0.06):
    """One diffusion step of the band boundaries (the second phase of csrc/mgpu.cpp rebalance()): a boundary moves one 16-row stripe towards the slower of
    the two ranks it separates when their times differ by more than `tol`; a band keeps at least one stripe; a rank whose band changed in this step is
    measured again before another of its boundaries moves, and the boundaries of the slowest ranks go first (then the larger imbalance).  A rank's time is not
    the sum of its stripes' costs (a band that holds horizon rows takes what its slowest tile takes), so the cost model of plan_bands settles with the slowest
    rank well above the fastest; from there the boundaries diffuse."""
    nb, n = list(part), len(part) - 1
    cand = []
    for k in range(1, n):
        a, b = max(1e-9, float(rank_ms[k - 1])), max(1e-9, float(rank_ms[k]))
        if max(a, b) > min(a, b) * (1.0 + tol):
            cand.append((-max(a, b), -max(a, b) / min(a, b), k))
    touched = [False] * n
    for _, _, k in sorted(cand):
        if touched[k - 1] or touched[k]:
            continue
        if rank_ms[k - 1] > rank_ms[k] and nb[k] - nb[k - 1] > 16:
            nb[k] -= 16; touched[k - 1] = touched[k] = True
        elif rank_ms[k] > rank_ms[k - 1] and nb[k + 1] - nb[k] > 16:
            nb[k] += 16; touched[k - 1] = touched[k] = True
    return nb


class LocalComm:
    """world == 1: every exchange is a no-op."""
    rank, world = 0, 1
    rx_bytes = dict.fromkeys(HALO_KINDS, 0)
    def all_gather_rows(self, tensor, pitch, part, async_op=False, kind="fallback"): return None
    def halo_exchange(self, items, async_op=False, kind="filter"): return []
    def gather_rows_to(self, tensor, pitch, part, dst=0, async_op=False): return None
    def any_flag(self, flag): return bool(flag)
    def all_gather_floats(self, values): return [list(values)]
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
        self._done = {}
        self.rx_bytes = dict.fromkeys(HALO_KINDS, 0)   # bytes this rank has received, by purpose (TiledFrame snapshots it per frame)
        # Test hook for the bench's tiled == untiled gate (bench.py verify_*, tests/test_tiled_gloo.py): the first row of every received filter-halo segment is
        # overwritten after it has arrived, so the gate must report a mismatch.  Never set outside tests.
        self._corrupt = os.environ.get("RESTIR_TEST_CORRUPT_HALO") == "1"
        self._rx_log = []

    # Every exchange is expressed on an arbitrary row partition `part` (world + 1 boundaries): rank q owns rows [part[q], part[q+1]).
    # A rank receives the rows it needs from whoever owns them and sends the rows others need from its own band — one batched
    # send/recv; both sides enumerate the segments of a pair in the same order (the receiver's need list), which is what NCCL's
    # in-order matching of a group requires.  Bands may be narrower than a halo (a halo then spans several ranks).
    def _batch(self, ops, unpack, async_op):
        host_sync = bool(ops) and not self.nccl and ops[0].tensor.is_cuda
        if host_sync:
            self.torch.cuda.synchronize()    # (gloo moving device tensors — bench.py's one-device test hook: the transport knows nothing of our streams)
        works = self.dist.batch_isend_irecv(ops) if ops else []
        p = Pending(self, [w for w in works if w is not None], unpack, keep=[op.tensor for op in ops], corrupt=self._rx_log)   # packed send buffers live until the exchange is waited for
        self._rx_log = []
        if not async_op:
            p.wait()
            return None
        return p

    def _view(self, item, a, b):
        return halo_view(item, a, b)

    def _pair_ops(self, ops, unpack, item, need_of, kind):
        """sends / receives of one item between this rank and every other; both sides enumerate the segments of a pair in the same order"""
        P2P, me, part = (self.dist.P2POp if self.dist is not None else None), self.rank, item.part
        packed = item.even or item.width is not None
        for q in range(self.world):
            if q == me:
                continue
            for (lo, hi) in need_of(me):                     # what I receive from q
                a, b = max(lo, part[q]), min(hi, part[q + 1], item.limit)
                v = self._view(item, a, b) if b > a else None
                if v is None:
                    continue
                self.rx_bytes[item.kind or kind] += item.nbytes(a, b)
                if P2P is None:
                    continue                                 # CountingComm: the accounting is all there is
                if self._corrupt and (item.kind or kind) == "filter":
                    self._rx_log.append(v)
                if packed:
                    stage = self.torch.empty(v.shape, dtype=v.dtype, device=v.device)
                    unpack.append((v, stage))
                    ops.append(P2P(self.dist.irecv, stage, q, group=self.group))
                else:
                    ops.append(P2P(self.dist.irecv, v.reshape(-1), q, group=self.group))
            for (lo, hi) in need_of(q):                      # what q receives from me
                a, b = max(lo, part[me]), min(hi, part[me + 1], item.limit)
                v = self._view(item, a, b) if b > a else None
                if v is None or P2P is None:
                    continue
                ops.append(P2P(self.dist.isend, v.contiguous() if packed else v.reshape(-1), q, group=self.group))

    def all_gather_rows(self, tensor, pitch, part, async_op=False, kind="fallback"):
        ops, unpack = [], []
        item = Halo(tensor, pitch, part, part[-1], part[-1])
        self._pair_ops(ops, unpack, item, lambda r: [(0, part[-1])], kind)
        return self._batch(ops, unpack, async_op)

    def halo_exchange(self, items, async_op=False, kind="filter"):
        """items: Halo objects.  For every item fill its rows above and below this rank's band from the ranks that own them, all items in ONE batched
        send/recv.  Returns a Pending (async_op) or None."""
        ops, unpack = [], []
        for it in items:
            if it.halo <= it.inner:
                continue
            self._pair_ops(ops, unpack, it, it.need, kind)
        return self._batch(ops, unpack, async_op)

    def gather_rows_to(self, tensor, pitch, part, dst=0, async_op=False):
        # NCCL has no in-place gather primitive; grouped send/recv to dst: only dst's links carry the traffic
        ops, unpack = [], []
        item = Halo(tensor, pitch, part, part[-1], part[-1])
        self._pair_ops(ops, unpack, item, lambda r: [(0, part[-1])] if r == dst else [], "gather")
        return self._batch(ops, unpack, async_op)

    def all_gather_floats(self, values):
        """every rank's list of floats, by rank (host side; used outside the timed region to plan the band heights)"""
        t = self.torch.tensor([float(v) for v in values], dtype=self.torch.float64, device="cuda" if self.nccl else "cpu")
        out = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t, group=self.group)
        return [[float(x) for x in o.cpu()] for o in out]

    def any_flag(self, flag):
        self._flag.fill_(1 if flag else 0)
        self.dist.all_reduce(self._flag, op=self.dist.ReduceOp.MAX, group=self.group)
        return bool(int(self._flag.item()))

    def wait(self, work):
        """NCCL: makes the CURRENT stream wait for the exchange (may be called once per consuming stream).  gloo: blocks the host."""
        if work is None:
            return
        for w in (work if isinstance(work, (list, tuple)) else [work]):
            if w is not None:
                w.wait()

    def _wait_works(self, works):
        for w in works:
            w.wait()

    def barrier(self):
        self.dist.barrier(group=self.group)


class CountingComm(TorchComm):
    """The exchange plan of rank `rank` of a `world`-rank job without any transport: every halo_exchange / all_gather is enumerated exactly as TorchComm does
    and only its received bytes are counted (rx_bytes).  With CountingBackend below this prices a frame's communication for any size / partition on a host
    without GPUs (steady_halo_bytes; tests/test_tiled_gloo.py holds it to the bytes a real gloo run receives)."""
    def __init__(self, rank, world):
        import torch
        self.torch, self.dist, self.group, self.rank, self.world, self.nccl = torch, None, None, rank, world, False
        self._done, self._corrupt, self._rx_log = {}, False, []
        self.rx_bytes = dict.fromkeys(HALO_KINDS, 0)
    def _batch(self, ops, unpack, async_op): return None
    def all_gather_floats(self, values): return [list(values) for _ in range(self.world)]
    def any_flag(self, flag): return bool(flag)
    def barrier(self): pass


class CountingBackend:
    """backend of a TiledFrame that renders nothing: buffers are shape-only (meta) tensors of the real sizes"""
    _ELEM = {abi.BUF_GBUFFER0: 16, abi.BUF_GBUFFER1: 16, abi.BUF_DIRECT_RESV0: 36, abi.BUF_DIRECT_RESV1: 36, abi.BUF_DIRECT_RESV_TEMP: 36, abi.BUF_LIGHT_ID0: 4,
             abi.BUF_LIGHT_ID1: 4, abi.BUF_INDIRECT_RESV0: 76, abi.BUF_INDIRECT_RESV1: 76}
    def __init__(self, width, height):
        import torch
        self.torch, self.W, self.H, self._c = torch, width, height, {}
    def run_stage(self, *a): pass
    def set_history_rows(self, r0, r1): pass
    def history_miss(self): return False
    def history_miss_stage(self, stage): return False
    def tensor(self, buf):
        if buf not in self._c:
            half = buf in (abi.BUF_INDIRECT_RESV0, abi.BUF_INDIRECT_RESV1)
            pitch = (self.W // 2 if half else self.W) * self._ELEM.get(buf, 16)
            self._c[buf] = (self.torch.empty(pitch * (self.H + 128), dtype=self.torch.uint8, device="meta"), pitch)
        return self._c[buf]


def steady_halo_bytes(width, height, world, rank, part=None, pipelined=True, denoise=1, restir=None, frames=3):
    """bytes rank `rank` receives per steady-state frame of a `world`-way row tiling, by purpose (no GPU, no process group)"""
    class _St:
        pass
    st = _St(); st.denoise = denoise; st.ReSTIRState = abi.RESTIR_TEMPORAL if restir is None else restir
    fr = (PipelinedTiledFrame if pipelined else TiledFrame)(CountingBackend(width, height), CountingComm(rank, world), width, height, part)
    fr._copy_state = lambda s_: s_
    for f in range(frames):
        fr.render_frame(st, f)
    return dict(fr.halo_bytes)


class TiledFrame:
    """Drives one backend (HIP Renderer or, in the CPU tests, the oracle) over this rank's row band.

    backend needs: run_stage(state, frames, stage, level, row_begin, row_end), tensor(buf) -> (flat uint8 torch tensor over the
    whole allocation, row pitch in bytes), set_history_rows(r0, r1), history_miss() -> bool (synchronising)."""

    def __init__(self, backend, comm, width, height, part=None):
        self.b, self.comm, self.W, self.H = backend, comm, width, height
        self.Hh = height // 2
        self._pending = []          # history halos in flight (sent at the end of the previous frame)
        self._result_pending = []
        self.history_fallbacks = 0  # frames that needed the full history (statistics)
        self._last_frames = None    # frame index of the last render_frame (set_partition redistributes its history)
        self._halo = HIST_HALO_MIN if HIST_HALO is None else int(HIST_HALO)   # history halo of the NEXT frame's temporal lookups (adaptive, see HIST_HALO)
        self._halo_cur = self._halo # ... and of the frame being rendered (its history rows were exchanged with this width)
        self._calm, self._fb_seen = 0, 0
        self.halo_bytes = dict.fromkeys(HALO_KINDS, 0)   # bytes this rank received for the last complete frame, by purpose
        self._rx_mark = dict(comm.rx_bytes)
        self._apply_partition(part if part is not None else equal_partition(height, comm.world))

    def _next_halo(self):
        """History halo of the next frame, decided once per frame from the fallbacks seen so far — identical on every rank (the miss flags are reduced over
        the ranks), same rule as csrc/mgpu.cpp."""
        self._halo_cur = self._halo
        if HIST_HALO is not None:
            self._halo = int(HIST_HALO)
        elif self.history_fallbacks != self._fb_seen:
            self._fb_seen, self._calm, self._halo = self.history_fallbacks, 0, min(HIST_HALO_MAX, max(HIST_HALO_MIN, self._halo * 2))
        else:
            self._calm += 1
            if self._calm >= HIST_HALO_CALM:
                self._calm, self._halo = 0, max(HIST_HALO_MIN, self._halo // 2)
        return self._halo

    def _account(self):
        """close the byte accounting of a frame"""
        now = self.comm.rx_bytes
        self.halo_bytes = {k: now[k] - self._rx_mark.get(k, 0) for k in HALO_KINDS}
        self._rx_mark = dict(now)

    def _gbuffer_halo(self, g, gp, hist):
        """the G-buffer rows the nine filter passes (and, up to `hist` rows, the next frame's temporal lookups) read beyond the band: every row close to the
        band, even rows only further out"""
        full = max(HALO_GBUFFER_FULL, hist)
        return [Halo(g, gp, self.part, full, self.H), Halo(g, gp, self.part, HALO_GBUFFER, self.H, inner=full, even=True)]

    def _apply_partition(self, part):
        part = [int(p) for p in part]
        if len(part) != self.comm.world + 1 or part[0] != 0 or part[-1] != self.H or any(p % 16 and p != self.H for p in part) or \
           any(part[i + 1] < part[i] for i in range(self.comm.world)):
            raise ValueError(f"bad row partition {part}: needs world+1 non-decreasing boundaries from 0 to H, multiples of 16 (trailing bands may be empty)")
        self.part, self.parth = part, half_partition(part, self.H)
        r = self.comm.rank
        self.y0, self.y1 = self.part[r], self.part[r + 1]
        self.h0, self.h1 = self.parth[r], self.parth[r + 1]

    def set_partition(self, part):
        """Switch to another row partition (e.g. cost-weighted band heights from plan_bands).  Everything in flight is finished and the
        history of the last frame is all-gathered under the OLD partition, so every rank holds the rows its new band + halo needs."""
        self.finish()
        if self.comm.world > 1 and self._last_frames is not None:
            cur = self._last_frames & 1
            # Both parities of the reservoir buffers: pixels that return early (miss, emitter, debug view) leave their reservoir slot
            # untouched (reference quirk), so a slot can hold the value of two frames ago — rows that change owner
            # must bring that along to stay bit-identical with the single-GPU frame.
            for buf, pt in ((abi.BUF_GBUFFER0 + cur, self.part), (abi.BUF_DIRECT_RESV0 + cur, self.part), (abi.BUF_LIGHT_ID0 + cur, self.part),
                            (abi.BUF_INDIRECT_RESV0 + cur, self.parth), (abi.BUF_DIRECT_RESV0 + (cur ^ 1), self.part),
                            (abi.BUF_LIGHT_ID0 + (cur ^ 1), self.part), (abi.BUF_INDIRECT_RESV0 + (cur ^ 1), self.parth),
                            (abi.BUF_DIRECT_RESV_TEMP, self.part)):
                t, p = self._t(buf)
                self.comm.all_gather_rows(t, p, pt, kind="moved")
            fn = getattr(self.b, "sync_all", None)
            if fn:
                fn()
        self._apply_partition(part)

    def rebalance(self, my_ms, smoothing=0.5, max_move=4):
        """One step of cost-weighted band planning (outside the timed region): `my_ms` is what this rank measured for its band (e.g. its
        direct + indirect stage launched alone); every rank contributes its number, the times are spread over the ranks' 16-row stripes
        into a smoothed per-stripe cost, plan_bands equalises the summed cost and set_partition moves the history.  Returns the new
        partition (identical on every rank: it is computed from the same gathered numbers)."""
        stripes = (self.H + 15) // 16
        if getattr(self, "_stripe_cost", None) is None or len(self._stripe_cost) != stripes:
            self._stripe_cost = [1.0] * stripes
        allms = self.comm.all_gather_floats([float(my_ms)])
        for q in range(self.comm.world):
            a, b = self.part[q] // 16, (self.part[q + 1] + 15) // 16
            if b <= a:
                continue
            per = max(1e-6, allms[q][0]) / (b - a)
            for s_ in range(a, b):
                self._stripe_cost[s_] = (1.0 - smoothing) * self._stripe_cost[s_] + smoothing * per
        new = plan_bands(self.H, self.comm.world, self._stripe_cost, prev=self.part, max_move=max_move)
        if new != self.part:
            self.set_partition(new)
        return new

    def diffuse(self, my_ms, tol=0.06):
        """One diffusion step (diffuse_bands) on the gathered per-rank times, after the cost-model rounds of rebalance(); every rank computes the same
        partition.  Returns (new partition, slowest / fastest of the gathered times)."""
        allms = [max(1e-6, a[0]) for a in self.comm.all_gather_floats([float(my_ms)])]
        new = diffuse_bands(self.part, allms, tol)
        if new != self.part:
            self.set_partition(new)
        return new, max(allms) / min(allms)

    def _t(self, buf):
        return self.b.tensor(buf)

    def _icol(self, frames):
        """The noisy indirect colour of frame `frames` (the HIP library keeps one buffer per frame parity: rt_select_frame)."""
        sel = getattr(self.b, "select_frame", None)
        if sel:
            sel(frames)
        return self._t(abi.BUF_DENOISE_IND_A)[0]

    def _run(self, state, frames, stage, level, r0, r1, limit):
        r0, r1 = max(0, r0), min(limit, r1)
        if r1 > r0:
            self.b.run_stage(state, frames, stage, level, r0, r1)

    def _direct(self, state, frames):
        """the direct stage on the band; with spatial reuse in two halves around an exchange of the cached reservoirs' boundary rows (the
        neighbour picks of direct_stage.comp:86-107 reach one pixel up / down): rt_run_stage levels 1 and 2"""
        if self.comm.world > 1 and state.ReSTIRState in (abi.RESTIR_SPATIAL, abi.RESTIR_SPATIOTEMPORAL):
            self._run(state, frames, abi.STAGE_DIRECT, 1, self.y0, self.y1, self.H)
            t, p = self._t(abi.BUF_DIRECT_RESV_TEMP)
            self.comm.halo_exchange([Halo(t, p, self.part, 2, self.H)], kind="spatial")
            self._run(state, frames, abi.STAGE_DIRECT, 2, self.y0, self.y1, self.H)
        else:
            self._run(state, frames, abi.STAGE_DIRECT, 0, self.y0, self.y1, self.H)

    def _traced_stages(self, state, frames):
        self._direct(state, frames)
        self._run(state, frames, abi.STAGE_INDIRECT, 0, self.h0, self.h1, self.Hh)

    def render_frame(self, state, frames):
        c, b = self.comm, self.b
        cur, last = frames & 1, (frames + 1) & 1
        single = c.world == 1
        self._last_frames = frames
        c.wait(self._pending)
        self._pending = []

        # ---- ray-traced stages on the band, with exact fallback when temporal reuse leaves band + history halo ------------
        if not single:
            b.set_history_rows(max(0, self.y0 - self._halo), min(self.H, self.y1 + self._halo))
        self._traced_stages(state, frames)
        if not single and c.any_flag(b.history_miss()):
            self.history_fallbacks += 1
            for buf, pt in ((abi.BUF_GBUFFER0 + last, self.part), (abi.BUF_DIRECT_RESV0 + last, self.part), (abi.BUF_LIGHT_ID0 + last, self.part),
                            (abi.BUF_INDIRECT_RESV0 + last, self.parth)):
                t, p = self._t(buf)
                c.all_gather_rows(t, p, pt, kind="fallback")
            b.set_history_rows(0, self.H)
            self._traced_stages(state, frames)
            b.history_miss()  # clear

        # ---- one neighbour exchange feeds all nine A-Trous passes --------------------------------------------------------
        pitch = self.W * _COLOR_BYTES
        if state.denoise > 0 and not single:
            g, gp = self._t(abi.BUF_GBUFFER0 + cur)
            dcol, _ = self._t(abi.BUF_DIRECT_RESULT0 + cur)
            icol = self._icol(frames)
            c.halo_exchange(self._gbuffer_halo(g, gp, 0) +
                            [Halo(dcol, pitch, self.part, HALO_DIRECT_COLOR, self.H),
                             Halo(icol, pitch, self.parth, HALO_INDIRECT_COLOR, self.Hh, width=(self.W // 2) * _COLOR_BYTES)], kind="filter")
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
            hh = self._next_halo()
            # (with the filters on, the rows of this frame's G-buffer within HALO_GBUFFER_FULL of the band are already here: the filter exchange brought them)
            g_have = HALO_GBUFFER_FULL if state.denoise > 0 else 0
            t, p = self._t(abi.BUF_GBUFFER0 + cur)
            items.append(Halo(t, p, self.part, hh, self.H, inner=min(g_have, hh)))
            for buf in (abi.BUF_DIRECT_RESV0 + cur, abi.BUF_LIGHT_ID0 + cur):
                t, p = self._t(buf)
                items.append(Halo(t, p, self.part, hh, self.H))
            t, p = self._t(abi.BUF_INDIRECT_RESV0 + cur)
            items.append(Halo(t, p, self.parth, hh // 2, self.Hh))
            self._pending = c.halo_exchange(items, async_op=True, kind="history")
            self._account()
            c.wait(self._result_pending)
            self._result_pending = []
            for buf in (abi.BUF_DIRECT_RESULT0 + cur, abi.BUF_INDIRECT_RESULT0 + cur):
                t, p = self._t(buf)
                w = c.gather_rows_to(t, p, self.part, dst=0, async_op=True)
                if w is not None:
                    self._result_pending.append(w)

    def finish(self):
        """Wait for everything in flight (call before reading results / at the end of a timed region)."""
        self.comm.wait(self._pending)
        self.comm.wait(self._result_pending)
        self._pending, self._result_pending = [], []


class _NullCtx:
    def __enter__(self): return self
    def __exit__(self, *a): return False


class PipelinedTiledFrame(TiledFrame):
    """TiledFrame with frames in flight on three streams per rank, the row-band version of rt_render_frame's overlap mode 2:

        main : direct(f)   | exchange X_D(f): G-buffer (144 rows), direct reservoirs + light ids (32), noisy direct colour (40)
        ind  : indirect(f) | exchange X_I(f): indirect reservoirs (16 half rows), noisy indirect colour (72 half rows)
        side : direct A-Trous x4 (f) ... indirect A-Trous x5 (f), compose(f), result bands -> rank 0

    direct(f+1) needs only X_D(f) — the neighbours' G-buffer / direct-reservoir history rows — so it starts while indirect(f) and
    the filters of frame f are still running.  A small band is latency bound (one multi-bounce tile is the critical path of the
    indirect stage however few rows the band has), so overlapping the stages is what makes more ranks pay off.  The second half
    of frame f (X_I, indirect A-Trous, compose, gather) is issued at the start of render_frame(f+1) after the host has seen the
    indirect stage's history-miss flag; finish() issues it for the last frame.

    Exactness: the G-buffer is triple buffered and the motion vectors double buffered (backend.rotate) because indirect(f)
    reads G(f) and G(f-1) while direct(f+1) writes; the history-miss flags are per stage kind (history_miss_stage); a miss
    drains everything and re-runs that stage with the all-gathered history, exactly like TiledFrame."""

    def __init__(self, backend, comm, width, height, part=None):
        super().__init__(backend, comm, width, height, part)
        self._wD = []            # X_D of the latest direct stage
        self._wD_prev = []       # ... and of the one before (history rows the indirect stage reads)
        self._rotated = None
        self._validated = True
        self._wI = []            # X_I of the latest finished frame
        self._wR = {}            # result gathers in flight, by frame parity
        self._prev = None        # (state copy, frames) whose second half is still to be issued
        self._ev = {}            # events: ("D"|"I"|"done", frames) -> event
        self._f0 = None

    # -- backend shims: the CPU test backends have no streams / events / rotation ------------------------------------------------
    def _stream(self, name):
        fn = getattr(self.b, "stream", None)
        return fn(name) if fn else _NullCtx()

    def _record(self, key):
        fn = getattr(self.b, "record", None)
        self._ev[key] = fn() if fn else None
        for k in [k for k in self._ev if k[1] < key[1] - 4]:
            del self._ev[k]

    def _wait_ev(self, key):
        ev = self._ev.get(key)
        if ev is not None:
            self.b.wait_event(ev)

    def _miss(self, stage):
        fn = getattr(self.b, "history_miss_stage", None)
        return fn(stage) if fn else self.b.history_miss()

    def _camera(self):
        r = getattr(self.b, "r", None) or getattr(self.b, "o", None)   # RendererTensors.r / the CPU test backend's oracle
        return getattr(r, "camera", None)

    def _set_camera(self, cam):
        r = getattr(self.b, "r", None) or getattr(self.b, "o", None)
        if cam is not None and r is not None:
            r.set_camera(cam)

    def _copy_state(self, state):
        return type(state).from_buffer_copy(state)

    def _drain(self):
        fn = getattr(self.b, "sync_all", None)
        if fn:
            fn()

    # -- frame -----------------------------------------------------------------------------------------------------------------
    def render_frame(self, state, frames):
        c, b = self.comm, self.b
        if c.world == 1 and os.environ.get("RESTIR_TILED_FORCE_PIPELINE") != "1":   # (forced: scripts/r06_host_period.py times THIS host's schedule on one band, no exchanges)
            return super().render_frame(state, frames)
        f, cur, last = frames, frames & 1, (frames + 1) & 1
        self._last_frames = frames
        self._rotate(f)
        # ---- 1. direct(f) on the main stream ---------------------------------------------------------------------------------
        with self._stream("main"):
            self._wait_ev(("done", f - 2)); self._wait_ev(("I", f - 2))     # storage of dcol(f) / G(f) is free again
            c.wait(self._wR.pop(cur, None))                                   # ... and the result band of f-2 has left
            c.wait(self._wD)                                                  # neighbours' history rows of f-1
            b.set_history_rows(max(0, self.y0 - self._halo), min(self.H, self.y1 + self._halo))
            self._direct(state, f)
            # ---- 2. second half of frame f-1, issued while direct(f) runs: the host's wait for indirect(f-1)'s flag overlaps
            #         with direct(f) instead of following it
            self._finish_prev()
            if c.any_flag(self._miss(abi.STAGE_DIRECT)):
                self.history_fallbacks += 1
                self._drain()
                for buf in (abi.BUF_GBUFFER0 + last, abi.BUF_DIRECT_RESV0 + last, abi.BUF_LIGHT_ID0 + last):
                    t, p = self._t(buf)
                    c.all_gather_rows(t, p, self.part, kind="fallback")
                b.set_history_rows(0, self.H)
                self._direct(state, f)
                self._miss(abi.STAGE_DIRECT)  # clear
            # the history halo of frame f+1 is decided here: every exchange that feeds its temporal lookups (X_D(f) below, X_I(f) in _finish_prev) uses it
            hh = self._next_halo()
            g, gp = self._t(abi.BUF_GBUFFER0 + cur)
            items = self._gbuffer_halo(g, gp, hh) if state.denoise > 0 else [Halo(g, gp, self.part, hh, self.H)]
            for buf in (abi.BUF_DIRECT_RESV0 + cur, abi.BUF_LIGHT_ID0 + cur):
                t, p = self._t(buf)
                items.append(Halo(t, p, self.part, hh, self.H, kind="history"))
            if state.denoise > 0:
                dcol, _ = self._t(abi.BUF_DIRECT_RESULT0 + cur)
                items.append(Halo(dcol, self.W * _COLOR_BYTES, self.part, HALO_DIRECT_COLOR, self.H))
            self._wD_prev = self._wD
            self._wD = c.halo_exchange(items, async_op=True, kind="history" if state.denoise == 0 else "filter")   # ONE batched send / recv per exchange
            self._record(("D", f))
        # ---- 3. indirect(f) on the ind stream -----------------------------------------------------------------------------------
        with self._stream("ind"):
            self._wait_ev(("D", f)); self._wait_ev(("done", f - 1))            # this G-buffer band; the noisy-indirect scratch is free
            c.wait(self._wD_prev); c.wait(self._wI)                           # G(f-1) / indirect-reservoir history rows
            b.set_history_rows(max(0, self.y0 - self._halo_cur), min(self.H, self.y1 + self._halo_cur))
            self._run(state, f, abi.STAGE_INDIRECT, 0, self.h0, self.h1, self.Hh)
            self._validated = False
            if getattr(b, "rotate", None) is None:
                # a backend without buffer rotation (the sequential CPU test backends) cannot re-run indirect(f) once
                # direct(f+1) has overwritten the G-buffer it reprojects into: validate right away
                self._validate_indirect(state, f)
            self._record(("I", f))
        # ---- 4. direct A-Trous(f) on the side stream ------------------------------------------------------------------------------
        with self._stream("side"):
            self._wait_ev(("D", f))
            c.wait(self._wD)
            if state.denoise > 0:
                for l in range(4):
                    self._run(state, f, abi.STAGE_DENOISE_DIRECT, l, self.y0 - DIRECT_GROW[l], self.y1 + DIRECT_GROW[l], self.H)
        self._prev = (self._copy_state(state), f, self._camera())
        self._account()

    def _finish_prev(self):
        if self._prev is None:
            return
        c, b = self.comm, self.b
        state, f, cam = self._prev
        self._prev = None
        cur, last = f & 1, (f + 1) & 1
        now = self._camera()
        self._set_camera(cam)           # the deferred launches belong to frame f: its camera, not the one set for f+1
        with self._stream("ind"):
            if not self._validated:
                self._validate_indirect(state, f)
            t, p = self._t(abi.BUF_INDIRECT_RESV0 + cur)
            items = [Halo(t, p, self.parth, self._halo // 2, self.Hh, kind="history")]
            if state.denoise > 0:
                icol = self._icol(f)
                items.append(Halo(icol, self.W * _COLOR_BYTES, self.parth, HALO_INDIRECT_COLOR, self.Hh, width=(self.W // 2) * _COLOR_BYTES))
            self._wI = c.halo_exchange(items, async_op=True, kind="filter")
            self._record(("Ix", f))
        with self._stream("side"):
            self._wait_ev(("Ix", f)); self._wait_ev(("I", f))
            c.wait(self._wI)
            if state.denoise > 0:
                for l in range(5):
                    self._run(state, f, abi.STAGE_DENOISE_INDIRECT, l, self.h0 - INDIRECT_GROW[l], self.h1 + INDIRECT_GROW[l], self.Hh)
            self._run(state, f, abi.STAGE_COMPOSE, 0, self.y0, self.y1, self.H)
            self._record(("done", f))
            works = []
            for buf in (abi.BUF_DIRECT_RESULT0 + cur, abi.BUF_INDIRECT_RESULT0 + cur):
                t, p = self._t(buf)
                w = c.gather_rows_to(t, p, self.part, dst=0, async_op=True)
                if w is not None:
                    works.append(w)
            self._wR[cur] = works
        self._set_camera(now)

    def _validate_indirect(self, state, f):
        """History-miss check of indirect(f) (waits for it), with the exact fallback: all-gather G(f-1) and the indirect
        reservoirs of f-1, re-run the stage.  Must be called with the ind stream current."""
        c, b = self.comm, self.b
        last = (f + 1) & 1
        self._validated = True
        if not c.any_flag(self._miss(abi.STAGE_INDIRECT)):
            return
        self.history_fallbacks += 1
        self._drain()
        newer = self._rotated
        if newer is not None and newer > f:
            self._rotate(newer)            # undo: the boundary ids point at frame f's G-buffers / motion vectors again
        for buf, pt in ((abi.BUF_GBUFFER0 + last, self.part), (abi.BUF_INDIRECT_RESV0 + last, self.parth)):
            t, p = self._t(buf)
            c.all_gather_rows(t, p, pt, kind="fallback")
        b.set_history_rows(0, self.H)
        self._run(state, f, abi.STAGE_INDIRECT, 0, self.h0, self.h1, self.Hh)
        self._miss(abi.STAGE_INDIRECT)     # clear
        self._drain()
        if newer is not None and newer > f:
            self._rotate(newer)            # redo

    def _rotate(self, frames):  # remembers which frame the boundary ids of the rotated buffers currently belong to
        fn = getattr(self.b, "rotate", None)
        if fn:
            fn(frames)
        self._rotated = frames

    def finish(self):
        self._finish_prev()
        with self._stream("main"):
            self.comm.wait(self._wD); self.comm.wait(self._wI)
            for w in self._wR.values():
                self.comm.wait(w)
        self._wR = {}
        self._drain()


class RendererTensors:
    """Backend adapter: HIP Renderer + torch views of its HBM buffers (zero copy through __cuda_array_interface__)."""
    def __init__(self, renderer):
        import torch
        self.r, self.torch = renderer, torch
        self._cache = {}
        self._streams = None
        # rt_create gives the context its own non-blocking stream; NCCL's work.wait() only orders torch's current stream.  Bind the
        # renderer to it here so that halo exchanges and all-gathers are ordered against the stage kernels even when the caller forgets
        renderer.set_stream(torch.cuda.current_stream().cuda_stream)
    def run_stage(self, state, frames, stage, level, r0, r1):
        self.r.run_stage(state, frames, stage, level, r0, r1)
    def set_history_rows(self, r0, r1):
        self.r.set_history_rows(r0, r1)
    def history_miss(self):
        return self.r.history_miss()
    # ---- frames in flight (PipelinedTiledFrame) ----
    def history_miss_stage(self, stage):
        return self.r.history_miss_stage(stage)
    def rotate(self, frames):
        self.r.rotate_buffers(frames)
        for buf in (abi.BUF_GBUFFER0, abi.BUF_GBUFFER1, abi.BUF_MOTION):
            self._cache.pop(buf, None)
    @staticmethod
    def create_streams(renderer):
        """The rank's three streams = the CONTEXT's (rt_get_streams, ABI 2.3): created on the first call, filter stream first, then the indirect stream, with the levels of
        RESTIR_TILED_PRIO (streams that get high priority: any of `ind`, `side`; default `ind`).  Round 5 found that a stream's worth depends on how many streams the
        process created before it; until round 6 this host took three streams from torch's pool (32 per priority class, created on first use — after RCCL's).  Call before
        torch.distributed.init_process_group.  Measured (per-rank emulation, N = 2 / 8, round 3): none 2.94 / 2.14 ms, "ind" 2.59 / 1.88, "ind,side" 2.60 / 1.90 — the
        indirect stage carries the critical path (one multi-bounce tile), its waves should not queue behind the next frame's direct stage."""
        import os
        prio = os.environ.get("RESTIR_TILED_PRIO", "ind").split(",")
        if renderer.stream_layout()["creation_index"]["ind"] < 0:
            renderer.set_stream_priorities(1 if "ind" in prio else 0, 1 if "side" in prio else 0)
        return renderer.streams()
    def stream(self, name):
        """Context: kernels (rt_set_stream) and collectives (torch current stream) issued inside go to the named stream."""
        if self._streams is None:
            import os
            if os.environ.get("RESTIR_TILED_TORCH_STREAMS") == "1":   # A/B only (scripts/r06_host_period.py): the layout of rounds 1-5, three streams of torch's pool
                prio = os.environ.get("RESTIR_TILED_PRIO", "ind").split(",")
                self._streams = {n: self.torch.cuda.Stream(priority=-1 if n in prio else 0) for n in ("main", "ind", "side")}
            else:
                ptrs = RendererTensors.create_streams(self.r)
                self._streams = {n: self.torch.cuda.ExternalStream(ptrs[n]) for n in ("main", "ind", "side")}
        backend, s = self, self._streams[name]
        class _Ctx:
            def __enter__(self_):
                self_.prev = backend.torch.cuda.current_stream()
                backend.torch.cuda.set_stream(s); backend.r.set_stream(s.cuda_stream)
            def __exit__(self_, *a):
                backend.torch.cuda.set_stream(self_.prev); backend.r.set_stream(self_.prev.cuda_stream)
                return False
        return _Ctx()
    def record(self):
        ev = self.torch.cuda.Event(); ev.record(self.torch.cuda.current_stream()); return ev
    def wait_event(self, ev):
        self.torch.cuda.current_stream().wait_event(ev)
    def sync_all(self):
        self.torch.cuda.synchronize()
    def select_frame(self, frames):
        self.r.select_frame(frames)
        self._ind_parity = frames & 1
    def tensor(self, buf):
        key = (buf, getattr(self, "_ind_parity", 0)) if buf == abi.BUF_DENOISE_IND_A else buf   # that buffer exists once per frame parity
        if key not in self._cache:
            arr, pitch = self.r.device_array(buf)
            self._cache[key] = (self.torch.as_tensor(arr, device=f"cuda:{self.torch.cuda.current_device()}"), pitch)
        return self._cache[key]
