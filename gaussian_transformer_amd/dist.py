"""One camera per GPU: data-parallel rendering with a single gradient all-reduce.

The reference is single-device (utils/general_utils.py:133 pins cuda:0; no torch.distributed
anywhere), so this is new functionality layered on its train step (train.py:84-93,113-128):
every rank holds a replica of the Gaussian parameters, renders its own camera of the batch,
back-propagates its own loss, and the parameter gradients of all cameras are summed with ONE
collective over a flat bucket:
    59 floats per Gaussian at M = 16 (xyz 3, f_dc 3, f_rest 3(M-1), opacity 1, scaling 3, rotation 4)
  +  2 floats per Gaussian of per-view densification statistics
       (|dL/dmean2D[:, :2]| * visible, visible)  -- scene/gaussian_model.py:405-407 is per view,
       so the norm is taken before the reduction, not after.
The screen-space radii (train.py:115) are max-reduced separately (4 bytes per Gaussian), which makes the density
control of densify.py rank-consistent: same statistics + an identically seeded split generator = same Gaussians.
One process per GPU (torch.distributed.run); backend "nccl" is RCCL over xGMI on ROCm, "gloo"
in the CPU tests.  No collective is issued inside the rasterizer itself.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import torch
import torch.distributed as dist


def direct_all_reduce(flat: torch.Tensor, group=None, scratch: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Sum `flat` over the ranks with two direct exchanges instead of a ring: every rank owns one 1/n slice,
    receives that slice from each peer at once (n-1 point-to-point transfers, one per xGMI link), sums, and sends the
    reduced slice back to every peer.  Per link and phase S/n bytes, against 2(n-1)/n S through the slowest link of a ring.
    The owner adds the n contributions in RANK ORDER (0, 1, ..., n-1, its own in its place), so the result of an element
    does not depend on which rank owns it, i.e. not on where the element sits in the buffer: a compacted (sparse) exchange
    gives bitwise the same sums as the dense one, and every rank holds identical bits.
    Point-to-point only (batch_isend_irecv), so it also runs on gloo.  In place; returns `flat`.
    Not the default of bench.py: which of the two is faster on an 8-GPU node has not been measured (one GPU per box here)."""
    world = dist.get_world_size(group)
    if world == 1:
        return flat
    rank = dist.get_rank(group)
    n = flat.numel()
    per = (n + world - 1) // world
    lo = lambda r: min(r * per, n)
    hi = lambda r: min((r + 1) * per, n)
    mine = flat[lo(rank):hi(rank)]
    if scratch is None or scratch.numel() < (world - 1) * per or scratch.device != flat.device:
        scratch = torch.empty(((world - 1) * per,), dtype=flat.dtype, device=flat.device)
    peers = [r for r in range(world) if r != rank]
    to_global = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    # phase 1: my slice of everyone's buffer comes to me
    ops, bufs = [], {}
    for k, r in enumerate(peers):
        buf = scratch[k * per:k * per + mine.numel()]
        bufs[r] = buf
        if mine.numel():
            ops.append(dist.P2POp(dist.irecv, buf, to_global(r), group))
        if hi(r) > lo(r):
            ops.append(dist.P2POp(dist.isend, flat[lo(r):hi(r)], to_global(r), group))
    for w in (dist.batch_isend_irecv(ops) if ops else []):
        w.wait()
    if mine.numel():
        if rank == 0:
            for r in peers:
                mine += bufs[r]
        else:                                   # canonical order: ((x0 + x1) + x2) + ...
            tot = bufs[0]
            for r in range(1, world):
                tot += (mine if r == rank else bufs[r])
            mine.copy_(tot)
    # phase 2: the reduced slices go back to everyone
    ops = []
    for r in peers:
        if hi(r) > lo(r):
            ops.append(dist.P2POp(dist.irecv, flat[lo(r):hi(r)], to_global(r), group))
        if mine.numel():
            ops.append(dist.P2POp(dist.isend, mine, to_global(r), group))
    for w in (dist.batch_isend_irecv(ops) if ops else []):
        w.wait()
    return flat


class GradientExchange:
    """The one exchange step of the data-parallel path: sums the parameter gradients of all ranks' cameras.

    The gradients live in flat float32 "arenas" laid out as rasterizer.gradient_arena fills them
    ([means3D 3P | shs 3MP | opacity P | scales 3P | rotations 4P]; gsr_backward writes them there directly, no packing
    copy).  What this class adds to one blocking all-reduce of 236 B/Gaussian (M = 16):

      * overlap   -- launch() enqueues the exchange of the arena just filled on a side stream (RCCL's own stream for
                     all_reduce(async_op=True), an explicit HIP stream for the direct exchange) and returns; the next
                     camera's forward + backward run meanwhile into the OTHER arena (double buffering).  wait() / finish()
                     make the compute stream wait, and must be called before the optimiser reads the gradients.  Within one
                     iteration that renders B cameras per rank this hides every exchange but the last one without delaying any
                     update; with B = 1 it is a one-step-delayed update and the caller must want that.
      * buckets   -- the arena is cut at parameter boundaries into pieces of at most `bucket_bytes`, one collective each,
                     so the first pieces are on the wire while the later ones are still queued (ring all-reduce is per-link
                     bound on the xGMI mesh: many medium collectives pipeline better than one 236 MB one).
      * active SH -- while the active SH degree D is below the stored one (train.py:72-73 raises it every 1000 iterations)
                     only the first (D+1)^2 coefficient columns of dL/dshs can be non-zero; `sh_active` exchanges just those
                     (3 instead of 48 floats per Gaussian at degree 0).
      * sparse    -- `launch(visible=mask)`: a Gaussian no camera composited has an all-zero gradient row.  The ranks first agree on
                     the union of their masks (one MAX all-reduce of P bytes), compact the rows of that union into a buffer of fixed
                     capacity (no host synchronisation: the union's size is read when the exchange is waited for), exchange it and
                     scatter the sums back (rows outside the union stay zero).  With algo="direct" the sums are bitwise those of
                     the dense exchange (see direct_all_reduce).  The very first sparse exchange is as large as the dense one.
    bytes_last: payload bytes this rank handed to the collective(s) in the last launch()."""

    def __init__(self, P: int, M: int, device, has_scale_rot: bool = True, mode: str = "overlap", algo: str = "allreduce",
                 bucket_bytes: int = 64 << 20, n_buffers: int = 2, group=None, sparse_slack=(1.25, 1024)):
        assert mode in ("sync", "overlap") and algo in ("allreduce", "direct")
        self.P, self.M, self.mode, self.algo, self.group = int(P), int(M), mode, algo, group
        self.device = torch.device(device)
        self.sizes = [3 * P, 3 * M * P, P] + ([3 * P, 4 * P] if has_scale_rot else [])
        self.names = ["means3D", "shs", "opacities"] + (["scales", "rotations"] if has_scale_rot else [])
        total = sum(self.sizes)
        self.arenas = [torch.zeros((total,), dtype=torch.float32, device=self.device) for _ in range(max(1, n_buffers if mode == "overlap" else 1))]
        self.bucket_floats = max(1, int(bucket_bytes) // 4)
        self.cur = 0
        self.pending = [None] * len(self.arenas)       # per arena: list of waitables of the exchange in flight
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.comm_stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self.scratch = None
        self.bytes_last = 0
        self.sh_active = None                            # None = all M coefficient columns
        self.sparse_slack = sparse_slack                 # capacity of the compacted buffer = last union * [0] + [1] rows
        self._union_known = None                         # size of the last union the sparse exchange looked at (None: never)
        self._count_host = None
        self._comp = None
        self.union_rows = 0
        self.sparse_overflows = 0

    # ---- buffers ----
    def arena(self) -> torch.Tensor:
        """The arena the NEXT backward pass should fill (its previous exchange is waited for first)."""
        self._wait_index(self.cur)
        return self.arenas[self.cur]

    def views(self, arena: Optional[torch.Tensor] = None):
        a = self.arenas[self.cur] if arena is None else arena
        out, off = {}, 0
        shapes = {"means3D": (self.P, 3), "shs": (self.P, self.M, 3), "opacities": (self.P, 1), "scales": (self.P, 3), "rotations": (self.P, 4)}
        for n, k in zip(self.names, self.sizes):
            out[n] = a[off:off + k].view(shapes[n]); off += k
        return out

    def _pieces(self, flat: torch.Tensor):
        """Contiguous pieces of `flat` to exchange: parameter boundaries, then at most bucket_floats each."""
        segs, off = [], 0
        for n, k in zip(self.names, self.sizes):
            segs.append((off, off + k)); off += k
        out = []
        for lo, hi in segs:
            while lo < hi:
                out.append(flat[lo:min(hi, lo + self.bucket_floats)]); lo += self.bucket_floats
        return out

    # ---- exchange ----
    def _reduce(self, t: torch.Tensor, waits: list):
        self.bytes_last += t.numel() * t.element_size()
        if self.algo == "direct":
            per = (t.numel() + self.world - 1) // self.world
            if self.scratch is None or self.scratch.numel() < (self.world - 1) * per or self.scratch.dtype != t.dtype:
                self.scratch = torch.empty(((self.world - 1) * per,), dtype=t.dtype, device=t.device)
            direct_all_reduce(t, self.group, self.scratch)
        elif self.mode == "overlap":
            waits.append(dist.all_reduce(t, group=self.group, async_op=True))
        else:
            dist.all_reduce(t, group=self.group)

    def launch(self, visible: Optional[torch.Tensor] = None) -> None:
        """Exchange the current arena (filled by the backward pass that just ran); switches to the next arena."""
        idx, flat = self.cur, self.arenas[self.cur]
        self.bytes_last = 0
        if self.world > 1:
            waits: list = []
            side = self.comm_stream is not None and self.mode == "overlap" and self.algo == "direct"
            if side:
                self.comm_stream.wait_stream(torch.cuda.current_stream(self.device))
            ctx = torch.cuda.stream(self.comm_stream) if side else _NullCtx()
            with ctx:
                if visible is not None:
                    self._launch_sparse(flat, visible, waits)
                elif self.sh_active is not None and self.sh_active < self.M:
                    self._launch_active_sh(flat, waits)
                else:
                    for piece in self._pieces(flat):
                        self._reduce(piece, waits)
                if side:
                    ev = torch.cuda.Event(); ev.record(self.comm_stream); waits.append(ev)
            self.pending[idx] = waits
            if self.mode == "sync":
                self._wait_index(idx)
        self.cur = (self.cur + 1) % len(self.arenas)

    def _launch_active_sh(self, flat, waits):
        v = self.views(flat)
        K = int(self.sh_active)
        for n in self.names:
            if n != "shs":
                for piece in self._pieces_of(v[n].reshape(-1)):
                    self._reduce(piece, waits)
        act = v["shs"][:, :K, :].contiguous()             # [P, K, 3]: the only columns that can be non-zero
        for piece in self._pieces_of(act.view(-1)):
            self._reduce(piece, waits)
        waits.append(_Deferred(lambda: v["shs"][:, :K, :].copy_(act)))

    def _pieces_of(self, t):
        return [t[lo:lo + self.bucket_floats] for lo in range(0, t.numel(), self.bucket_floats)]

    # rows the compacted buffer of the sparse exchange holds: the union of the last step that was looked at, plus a quarter
    def _sparse_capacity(self) -> int:
        if self._union_known is None:
            return self.P
        return min(self.P, int(self._union_known * self.sparse_slack[0]) + int(self.sparse_slack[1]))

    def _launch_sparse(self, flat, visible, waits):
        """Exchange only the rows of the union of the ranks' masks, WITHOUT a host synchronisation on the compute stream: the rows are
        compacted into a buffer of fixed capacity (what the union needed recently, + 25 %) through a prefix sum of the mask; the union's
        size goes to the host asynchronously and is looked at when the exchange is waited for -- a union that did not fit (the camera
        moved a lot) is exchanged densely then, from the gradients that are still untouched in the arena."""
        vis = visible.to(torch.uint8, copy=True).contiguous()          # a copy: the reduction below must not overwrite the caller's mask
        self.bytes_last += vis.numel()
        dist.all_reduce(vis, op=dist.ReduceOp.MAX, group=self.group)      # union of the ranks' visibility masks
        pos = torch.cumsum(vis, 0, dtype=torch.int64) - 1
        cap = self._sparse_capacity()
        count_dev = pos[-1:] + 1
        if self.device.type == "cuda":
            if self._count_host is None:
                self._count_host = torch.zeros((1,), dtype=torch.int64).pin_memory()
            self._count_host.copy_(count_dev, non_blocking=True)
            count_ev = torch.cuda.Event(); count_ev.record(torch.cuda.current_stream(self.device))
        else:
            self._count_host, count_ev = count_dev.clone(), None
        inside = (vis != 0) & (pos < cap)
        idx = torch.where(inside, pos, torch.full_like(pos, cap))         # everything else lands in the dump row `cap`
        v = self.views(flat)
        widths = [v[n].reshape(self.P, -1).shape[1] for n in self.names]
        W = sum(widths)
        if self._comp is None or self._comp.shape[0] < cap + 1 or self._comp.shape[1] != W:
            self._comp = torch.empty((cap + 1, W), dtype=flat.dtype, device=flat.device)
        comp = self._comp[:cap + 1]
        off = 0
        for n, w in zip(self.names, widths):                              # gathered per parameter straight into the compacted buffer
            comp[:, off:off + w].index_copy_(0, idx, v[n].reshape(self.P, w))
            off += w
        comp[cap].zero_()
        payload = comp[:cap]
        for piece in self._pieces_of(payload.reshape(-1)):
            self._reduce(piece, waits)

        def finish():
            if count_ev is not None:
                count_ev.synchronize()                                    # recorded before the exchange was even launched: long done
            n_union = int(self._count_host.item())
            self._union_known = n_union
            self.union_rows = n_union
            if n_union > cap:                                             # did not fit: the arena still holds this rank's own gradients
                self.sparse_overflows += 1
                late: list = []
                for piece in self._pieces(flat):
                    self._reduce(piece, late)                             # same algorithm (and summation order) as the dense exchange
                for w in late:
                    w.wait()
                return
            comp[cap].zero_()
            off2 = 0
            for n, w in zip(self.names, widths):                          # rows outside the union are zero on every rank: the dump row
                v[n].reshape(self.P, w).copy_(comp.index_select(0, idx)[:, off2:off2 + w])
                off2 += w
        waits.append(_Deferred(finish))

    def _wait_index(self, idx: int) -> None:
        waits = self.pending[idx]
        if not waits:
            return
        for w in waits:                                   # first the transfers ...
            if isinstance(w, _Deferred):
                continue
            if self.device.type == "cuda" and isinstance(w, torch.cuda.Event):
                torch.cuda.current_stream(self.device).wait_event(w)
            else:
                w.wait()                                  # c10d Work: the current stream waits for the collective
        for w in waits:                                   # ... then what consumes them (copy-back, scatter), on the compute stream
            if isinstance(w, _Deferred):
                w.run()
        self.pending[idx] = None

    def wait(self, arena_index: Optional[int] = None) -> None:
        """Make the compute stream wait for the exchange of one arena (default: the one launched last)."""
        self._wait_index((self.cur - 1) % len(self.arenas) if arena_index is None else arena_index)

    def finish(self) -> None:
        for i in range(len(self.arenas)):
            self._wait_index(i)


class _Deferred:
    """Work to run on the compute stream once the collectives queued before it are waited for (copy-back, scatter)."""

    def __init__(self, fn):
        self.fn = fn

    def run(self):
        self.fn()


class _NullCtx:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


class GradientBucket:
    """Flat, reusable float32 buffer holding every parameter gradient + the densification stats."""

    def __init__(self, params: Sequence[torch.Tensor], num_points: int):
        self.shapes = [tuple(p.shape) for p in params]
        self.sizes = [int(p.numel()) for p in params]
        self.P = int(num_points)
        total = sum(self.sizes) + 2 * self.P
        dev = params[0].device
        self.flat = torch.zeros((total,), dtype=torch.float32, device=dev)

    def pack(self, params: Sequence[torch.Tensor], grad_norm_vis: torch.Tensor, vis: torch.Tensor) -> None:
        off = 0
        for p, n in zip(params, self.sizes):
            g = p.grad
            if g is None:
                self.flat[off:off + n].zero_()
            else:
                self.flat[off:off + n].copy_(g.reshape(-1))
            off += n
        self.flat[off:off + self.P].copy_(grad_norm_vis.reshape(-1)); off += self.P
        self.flat[off:off + self.P].copy_(vis.reshape(-1).to(torch.float32))

    def unpack(self, params: Sequence[torch.Tensor], scale: float = 1.0):
        off = 0
        for p, n, shp in zip(params, self.sizes, self.shapes):
            g = self.flat[off:off + n].view(shp)
            if p.grad is None:
                p.grad = (g * scale).clone()
            else:
                p.grad.copy_(g * scale)
            off += n
        gnorm = self.flat[off:off + self.P].clone(); off += self.P
        count = self.flat[off:off + self.P].clone()
        return gnorm, count

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * 4


def data_parallel_step(pc, camera, pipe, bg: torch.Tensor, gt_image: Optional[torch.Tensor],
                       render_fn: Callable, loss_fn: Callable, bucket: Optional[GradientBucket] = None,
                       group=None, average: bool = True, dL_dimage: Optional[torch.Tensor] = None):
    """One data-parallel render step on this rank's camera.

    Mirrors train.py:86-93 (render -> loss -> backward) and :113-116 (densification statistics),
    then reduces.  Returns dict(loss, render, radii, visibility_filter, bucket).
    If dL_dimage is given it is used as the upstream gradient instead of a loss (bench mode).
    """
    params = pc.parameters()
    for p in params:
        p.grad = None
    pkg = render_fn(camera, pc, pipe, bg)
    image, vsp, vis, radii = pkg["render"], pkg["viewspace_points"], pkg["visibility_filter"], pkg["radii"]
    if dL_dimage is not None:
        loss = (image * dL_dimage).sum()
    else:
        loss = loss_fn(image, gt_image)
    loss.backward()
    P = pc.get_xyz.shape[0]
    if bucket is None:
        bucket = GradientBucket(params, P)
    g2 = vsp.grad if vsp.grad is not None else torch.zeros_like(vsp)
    gnorm = torch.norm(g2[:, :2], dim=-1) * vis.to(g2.dtype)
    bucket.pack(params, gnorm, vis)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world > 1:
        dist.all_reduce(bucket.flat, op=dist.ReduceOp.SUM, group=group)
    gnorm_sum, vis_count = bucket.unpack(params, scale=(1.0 / world) if average else 1.0)
    # per-view statistics, summed over the views of the batch (add_densification_stats once per view)
    if getattr(pc, "xyz_gradient_accum", None) is not None:
        pc.xyz_gradient_accum += gnorm_sum[:, None]
        pc.denom += vis_count[:, None]
    # largest screen-space radius over the views of the batch (train.py:115), so that every rank prunes the same
    # Gaussians afterwards (densify.py: seed the split generator identically on all ranks)
    if getattr(pc, "max_radii2D", None) is not None:
        r = torch.where(vis, radii.to(pc.max_radii2D.dtype), torch.zeros_like(pc.max_radii2D))
        if world > 1:
            dist.all_reduce(r, op=dist.ReduceOp.MAX, group=group)
        pc.max_radii2D = torch.max(pc.max_radii2D, r)
    return {"loss": loss.detach(), "render": image.detach(), "radii": radii, "visibility_filter": vis, "bucket": bucket}
