"""Spatial tiling of the vote_instances path: z-slabs with patch-radius halos.

One mechanism serves two purposes

* **multi-GPU** (one process per GPU, ``torch.distributed`` / RCCL): every rank owns a
  contiguous range of z-slabs of ONE volume, holds the prediction for its range grown by the
  halo, and the ranks meet in four collectives (scores, patch bits of the cover candidates,
  pair affinities, painted instance slabs);
* **single-GPU tiling** of volumes whose consensus array does not fit: the slabs of one rank
  are processed one after the other, the consensus living only for one slab at a time.

It replaces the reference's scale-out mechanism -- ``stitch_patch_graph.py``: blocks of
``chunksize`` + margin, a patch graph per block, inter-block edges recomputed on face overlaps,
one global labelling (:110-399, :553-669) -- but, unlike it, reproduces the WHOLE-VOLUME result
exactly: the order-defined global stages (ranking, greedy cover, thinning, component order) run
on globally gathered data, identically on every rank; only the voxel-local stages (S1 consensus,
S2 scores, S5 pair affinities, painting) are sharded.

A z-slab can further be cut into y/x tiles (``_yx_tiles=(ny, nx)``; single-process tiling of
volumes whose one-slice-thick slab would still not fit, e.g. 512^3 with 9^3 patches).  The same
halo rules hold per axis, except that y and x need the "pairs" growth on BOTH sides: a stored
consensus offset is lexicographically positive, i.e. its z component is >= 0 but its y / x
components take either sign, so the earlier voxel of a pair can lie on the high y / x side.

Halos along z (rz = pz // 2):
  scores of centres   [z0, z1)            need consensus bases [z0 - rz, z1 + rz)
  pairs with A in     [z0, z1)            need consensus bases [z0 - rz - (pz-1), z1 + rz)
  consensus of bases  [b0, b1)            needs the prediction on [b0 - 2 rz, b1 + 2 rz)
  foreground bits of partner patches B    need the prediction on [z0 - 2 pz - rz, z1 + 2 pz + rz)
so a rank holds its slabs grown by H = 2 pz + rz slices (clipped to the volume).  Local buffers
use local coordinates; ``ppp_params.origin_z`` carries the global offset (the per-pair LCG seed
of the patch-graph kernel is the only thing that depends on absolute coordinates).
"""
import logging
import ctypes
import os
import zlib

import numpy as np

from . import backend

logger = logging.getLogger(__name__)


def halo(patchshape):
    pz = int(patchshape[0])
    return 2 * pz + pz // 2


def plan_slabs(Z, n_slabs):
    """n_slabs contiguous z-ranges covering [0, Z)."""
    n_slabs = max(1, min(int(n_slabs), int(Z)))
    edges = np.linspace(0, Z, n_slabs + 1).round().astype(int)
    return [(int(edges[i]), int(edges[i + 1])) for i in range(n_slabs) if edges[i + 1] > edges[i]]


def plan_yx(Y, X, ny, nx):
    """ny x nx tiles (y0, y1, x0, x1) covering [0, Y) x [0, X)."""
    ys, xs = plan_slabs(Y, ny), plan_slabs(X, nx)
    return [(y0, y1, x0, x1) for (y0, y1) in ys for (x0, x1) in xs]


def slabs_of_rank(slabs, rank, world):
    """Contiguous block of slabs owned by `rank`."""
    per = [len(slabs) * r // world for r in range(world + 1)]
    return slabs[per[rank]:per[rank + 1]]


def local_range(my_slabs, Z, patchshape):
    """Global z-range [lo, hi) of the prediction a rank must hold for its slabs."""
    H = halo(patchshape)
    return max(0, my_slabs[0][0] - H), min(Z, my_slabs[-1][1] + H)


# ------------------------------------------------------------------------------------------
# communication
# ------------------------------------------------------------------------------------------
class LocalComm:
    """Single process."""
    rank, world = 0, 1

    def all_reduce_sum(self, t):
        return t

    def all_reduce_min(self, t):
        return t

    def all_reduce_max(self, t):
        return t

    def all_gather(self, t):
        return t.reshape((1,) + tuple(t.shape))

    def all_gather_slabs(self, vol, ranges):
        return vol

    def sendrecv(self, sends, recvs):
        if sends or recvs:
            raise ValueError("a single process has no peer")


def exchange_halo(own, own_range, need_range, comm, z_axis=0, chunk_bytes=1 << 30, out=None):
    """The halo exchange of the north star ("patch-radius halo exchanged over RCCL/xGMI"): a rank
    holds the slices [own_range) of a volume along `z_axis` -- its part of a RESIDENT prediction
    (C, z, Y, X), the U-Net's output as it stands on the GPU, or of a per-voxel field (z, Y, X) --
    and needs [need_range) (its slabs grown by halo(patchshape), clipped).  Every rank's two ranges
    are all-gathered; each pair of ranks moves the slices one owns and the other needs point to
    point (`comm.sendrecv`: batch_isend_irecv -- xGMI links are pairwise, a halo concerns the two
    or three ranks next to a boundary), `chunk_bytes` of the leading axes at a time (a z-range of
    a channel-major block is strided: the staging copy stays bounded).  Returns the tensor of
    [need_range) -- `own` itself when nothing is missing anywhere (no collective beyond the range
    gather).  out: a tensor of [need_range) to receive into; when `own` is the view of its own
    slices the own part is not copied -- a producer that writes its slab into the middle of a
    halo-sized buffer pays for the halo only (`_refresh_halo` of tiling.assemble).
    Reference analogue: block + margin loading, stitch_patch_graph.py:553-669."""
    import torch
    a, b = [int(v) for v in own_range]
    na, nb = [int(v) for v in need_range]
    if int(own.shape[z_axis]) != b - a:
        raise ValueError("`own` holds %d slices, own_range says %d" % (int(own.shape[z_axis]), b - a))
    if comm.world == 1:
        if (na, nb) != (a, b):
            raise ValueError("a single rank must hold all it needs")
        return own
    mine = torch.tensor([a, b, na, nb], dtype=torch.int64, device=own.device)
    ranges = [tuple(int(v) for v in r) for r in comm.all_gather(mine).cpu().numpy()]
    if all(r[2] >= r[0] and r[3] <= r[1] for r in ranges):
        return own
    if na > a or nb < b:
        raise ValueError("need_range must contain own_range")
    full_shape = list(own.shape)
    full_shape[z_axis] = nb - na
    if out is not None:
        if list(out.shape) != full_shape or out.dtype != own.dtype:
            raise ValueError("exchange_halo: `out` must hold [need_range)")
        full = out
        mid = full.narrow(z_axis, a - na, b - a)
        if mid.data_ptr() != own.data_ptr() or mid.stride() != own.stride():
            mid.copy_(own)
    else:
        full = torch.empty(full_shape, dtype=own.dtype, device=own.device)
        full.narrow(z_axis, a - na, b - a).copy_(own)
    covered = b - a
    # (peer, global z0, z1) of what I send / receive, every rank deriving the same lists
    sends, recvs = [], []
    for r, (ra, rb, rna, rnb) in enumerate(ranges):
        if r == comm.rank:
            continue
        for (lo, hi) in ((rna, min(rnb, ra)), (max(rna, rb), rnb)):       # what r misses, below / above its own
            s0, s1 = max(lo, a), min(hi, b)
            if s0 < s1:
                sends.append((r, s0, s1))
        for (lo, hi) in ((na, min(nb, a)), (max(na, b), nb)):             # what I miss ...
            s0, s1 = max(lo, ra), min(hi, rb)                              # ... and r owns
            if s0 < s1:
                recvs.append((r, s0, s1))
                covered += s1 - s0
    if covered != nb - na:
        raise ValueError("the ranks' own ranges do not cover [%d, %d)" % (na, nb))
    lead = int(np.prod(own.shape[:z_axis])) if z_axis else 1
    tail_bytes = int(np.prod(own.shape[z_axis + 1:])) * own.element_size()
    # the thickest piece ANY pair of ranks moves: every rank cuts the channel axis into the same steps
    thick = 1
    for (ra, rb, _, _) in ranges:
        for (_, _, qna, qnb) in ranges:
            thick = max(thick, min(qnb, rb) - max(qna, ra))
    thick = min(thick, max(rb - ra for ra, rb, _, _ in ranges))
    if z_axis == 0:
        steps = [(0, 1)]
    else:
        if z_axis != 1:
            raise NotImplementedError("exchange_halo: z_axis 0 or 1")
        per = max(1, int(chunk_bytes // max(1, thick * tail_bytes)))
        steps = [(c0, min(lead, c0 + per)) for c0 in range(0, lead, per)]
    moved = 0
    for (c0, c1) in steps:
        def cut(t, z0, z1, base):
            v = t.narrow(z_axis, z0 - base, z1 - z0)
            return v if z_axis == 0 else v[c0:c1]
        out = [(r, cut(own, s0, s1, a).contiguous()) for r, s0, s1 in sends]
        inn = [(r, torch.empty(cut(full, s0, s1, na).shape, dtype=own.dtype, device=own.device)) for r, s0, s1 in recvs]
        comm.sendrecv(out, inn)
        for (r, s0, s1), (_, buf) in zip(recvs, inn):
            cut(full, s0, s1, na).copy_(buf)
            moved += buf.numel() * buf.element_size()
    backend.note_add("halo_exchange_bytes_received", moved)
    return full


def gather_lists(comm, t):
    """[t of rank 0, t of rank 1, ...] for 1-d (or [n, k]) tensors whose length differs between
    the ranks: lengths first, then one all-gather of the tensors padded to the longest."""
    import torch
    if comm.world == 1:
        return [t]
    n = torch.tensor([int(t.shape[0])], dtype=torch.int64, device=t.device)
    lens = [int(v) for v in comm.all_gather(n).reshape(-1).cpu()]
    m = max(lens)
    if m == 0:
        return [t[:0] for _ in lens]
    pad = torch.zeros((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[:t.shape[0]] = t
    allp = comm.all_gather(pad)
    return [allp[r, :lens[r]] for r in range(comm.world)]


def rows_by_tile(rows, tiles):
    """{tile number: indices (ascending) of the pair rows whose patch A lies in that tile} in ONE
    pass over the list -- tile number of A per row (the tiles of a rank are a grid: per-axis
    position among the sorted cuts), one stable sort by it, a slice per tile -- instead of a scan of
    the whole list for every tile (6.8 s of 640 scans over 319 M rows at 1024^3).  None when the
    tiles are not a gap-free grid (the caller scans then)."""
    import torch
    dev = rows.device
    n_rows = int(rows.shape[0])
    cuts = [sorted(set(t[2 * a] for t in tiles)) for a in range(3)]
    tile_ends = [max(t[2 * a + 1] for t in tiles) for a in range(3)]
    grid_index = {}
    for n_t, t in enumerate(tiles):
        cell = 0
        for a in range(3):
            nxt = cuts[a].index(t[2 * a]) + 1
            if t[2 * a + 1] != (cuts[a][nxt] if nxt < len(cuts[a]) else tile_ends[a]):
                return None
            cell = cell * len(cuts[a]) + nxt - 1
        grid_index[cell] = n_t
    if len(grid_index) != len(tiles):
        return None
    tid = torch.zeros((n_rows,), dtype=torch.int64, device=dev)
    inside = torch.ones((n_rows,), dtype=torch.bool, device=dev)
    for a in range(3):
        coord = rows[:, a].to(torch.int64)
        bounds = torch.tensor(cuts[a], dtype=torch.int64, device=dev)
        pos = torch.searchsorted(bounds, coord, right=True) - 1
        inside &= (pos >= 0) & (coord < tile_ends[a])
        tid = tid * len(cuts[a]) + pos.clamp_(min=0)
        del coord, pos, bounds
    tid[~inside] = -1
    order = torch.argsort(tid, stable=True)
    sorted_tid = tid[order]
    cells = sorted(grid_index)
    keys = torch.tensor(cells, dtype=torch.int64, device=dev)
    seg_a = torch.searchsorted(sorted_tid, keys, right=False).cpu().tolist()
    seg_b = torch.searchsorted(sorted_tid, keys, right=True).cpu().tolist()
    return {grid_index[c]: order[a:b] for c, a, b in zip(cells, seg_a, seg_b)}


class TorchDistComm:
    """torch.distributed (backend "nccl" = RCCL over xGMI on the GPUs, "gloo" on CPU)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        # gloo moves device tensors through the host (tests: several ranks sharing one GPU)
        self.via_host = dist.get_backend(group) == "gloo"

    # element types RCCL reduces / moves (torch's ProcessGroupNCCL maps no 16-bit integer type);
    # anything else travels as int32 (reductions) or as raw bytes (gathers, point to point)
    _WIRE_OK = ("torch.int8", "torch.uint8", "torch.int32", "torch.int64", "torch.float16",
                "torch.bfloat16", "torch.float32", "torch.float64")

    def _reduce(self, t, op):
        if self.world > 1 and str(t.dtype) not in self._WIRE_OK:
            import torch
            wide = t.to(torch.int32)
            self._reduce(wide, op)
            t.copy_(wide.to(t.dtype))
            return t
        if self.world > 1:
            if self.via_host and t.is_cuda:
                h = t.cpu()
                self.dist.all_reduce(h, op=op, group=self.group)
                t.copy_(h)
            elif not self.via_host and not t.is_cuda:
                d = t.cuda()                     # (RCCL only moves device memory)
                self.dist.all_reduce(d, op=op, group=self.group)
                t.copy_(d)
            else:
                self.dist.all_reduce(t, op=op, group=self.group)
        return t

    def all_reduce_sum(self, t):
        """In-place sum over ranks.  Every element is non-zero on at most one rank (each voxel,
        candidate or pair row has one owner), so the sum is an exact gather."""
        return self._reduce(t, self.dist.ReduceOp.SUM)

    def all_reduce_min(self, t):
        return self._reduce(t, self.dist.ReduceOp.MIN)

    def all_reduce_max(self, t):
        return self._reduce(t, self.dist.ReduceOp.MAX)

    def all_gather(self, t):
        """[world, ...] stack of every rank's `t` (same shape everywhere)."""
        import torch
        if str(t.dtype) not in self._WIRE_OK:
            # a gather is type-agnostic: send the bytes
            raw = self.all_gather(t.contiguous().view(torch.uint8))
            return raw.view(t.dtype).reshape((self.world,) + tuple(t.shape))
        if self.via_host and t.is_cuda:
            src = t.cpu()
        elif not self.via_host and not t.is_cuda:
            src = t.cuda()                       # (RCCL only moves device memory)
        else:
            src = t.contiguous()
        parts = [torch.empty_like(src) for _ in range(self.world)]
        self.dist.all_gather(parts, src, group=self.group)
        return torch.stack(parts, 0).to(t.device)

    def all_gather_slabs(self, vol, ranges):
        """vol (Z, ...) holds valid data on this rank's z-range only; ranges = every rank's
        (z0, z1) in rank order.  One all-gather of the owned slabs (padded to the thickest one)
        completes it on every rank: each rank sends its share once, instead of the full-volume
        SUM all-reduce that moves ~2x the volume per rank and adds zeros."""
        import torch
        if self.world == 1:
            return vol
        if vol.dtype == torch.int16:
            # neither RCCL nor gloo moves 16-bit integers: a gather is type-agnostic, send bytes
            self.all_gather_slabs(vol.view(torch.uint8), ranges)
            return vol
        zmax = max(b - a for a, b in ranges)
        a, b = ranges[self.rank]
        send = torch.zeros((zmax,) + tuple(vol.shape[1:]), dtype=vol.dtype, device=vol.device)
        send[:b - a] = vol[a:b]
        if self.via_host and send.is_cuda:
            send_h = send.cpu()
            parts = [torch.empty_like(send_h) for _ in range(self.world)]
            self.dist.all_gather(parts, send_h, group=self.group)
            for r, (ra, rb) in enumerate(ranges):
                if r != self.rank:
                    vol[ra:rb] = parts[r][:rb - ra].to(vol.device)
            return vol
        recv = torch.empty((self.world * zmax,) + tuple(send.shape[1:]), dtype=vol.dtype, device=vol.device)
        self.dist.all_gather_into_tensor(recv, send, group=self.group)     # concatenated along z
        recv = recv.view((self.world, zmax) + tuple(send.shape[1:]))
        for r, (ra, rb) in enumerate(ranges):
            if r != self.rank:
                vol[ra:rb] = recv[r, :rb - ra]
        return vol

    def sendrecv(self, sends, recvs):
        """sends / recvs: [(peer rank, contiguous tensor)] -- one batch of point-to-point transfers
        (batch_isend_irecv; over RCCL a grouped ncclSend / ncclRecv on the xGMI link of each pair).
        Both sides order a pair's messages by ascending global position, which the callers
        guarantee by building both lists from the same all-gathered ranges."""
        import torch
        if not sends and not recvs:
            return
        if any(str(t.dtype) not in self._WIRE_OK for _, t in list(sends) + list(recvs)):
            as_bytes = lambda items: [(p, t.view(torch.uint8)) for p, t in items]       # noqa: E731
            return self.sendrecv(as_bytes(sends), as_bytes(recvs))
        host = self.via_host and any(t.is_cuda for _, t in list(sends) + list(recvs))
        to_dev = (not self.via_host) and any(not t.is_cuda for _, t in list(sends) + list(recvs))
        conv = (lambda t: t.cpu()) if host else ((lambda t: t.cuda()) if to_dev else (lambda t: t))
        s_buf = [(p, conv(t)) for p, t in sends]
        r_buf = [(p, torch.empty_like(conv(t))) if (host or to_dev) else (p, t) for p, t in recvs]
        ops = [self.dist.P2POp(self.dist.isend, t, p, self.group) for p, t in s_buf]
        ops += [self.dist.P2POp(self.dist.irecv, t, p, self.group) for p, t in r_buf]
        for w in self.dist.batch_isend_irecv(ops):
            w.wait()
        if host or to_dev:
            for (_, dst), (_, src) in zip(recvs, r_buf):
                dst.copy_(src.to(dst.device))

    def neighbour_min(self, items):
        """items: [(peer rank, tensor)].  Every tensor is replaced by the element-wise minimum
        of this rank's and the peer's tensor of the same position in the peer's list for us
        (point-to-point: xGMI links are pairwise, and a slab boundary concerns two ranks -- an
        all-reduce over all ranks moves every boundary's buffer to everybody)."""
        import torch
        if self.world == 1 or not items:
            return
        host = self.via_host and items[0][1].is_cuda
        if any(str(t.dtype) not in self._WIRE_OK for _, t in items):
            wide = [(p, t.to(torch.int32)) for p, t in items]
            self.neighbour_min(wide)
            for (_, t), (_, w) in zip(items, wide):
                t.copy_(w.to(t.dtype))
            return
        mine = [(p, (t.cpu() if host else t.contiguous())) for p, t in items]
        recv = [torch.empty_like(t) for _, t in mine]
        ops = []
        # lower peer first on both sides of a boundary: matching order of sends and receives
        for k in sorted(range(len(mine)), key=lambda k: mine[k][0]):
            p, t = mine[k]
            ops.append(self.dist.P2POp(self.dist.isend, t, p, self.group))
            ops.append(self.dist.P2POp(self.dist.irecv, recv[k], p, self.group))
        for w in self.dist.batch_isend_irecv(ops):
            w.wait()
        for (p, t_orig), (_, t), r in zip(items, mine, recv):
            m = torch.minimum(t, r)
            t_orig.copy_(m.to(t_orig.device) if host else m)


# ------------------------------------------------------------------------------------------
# prediction providers
# ------------------------------------------------------------------------------------------
class ZarrProvider:
    """Prediction provider over a stored ``volumes/pred_affs`` array (C, Z, Y, X) -- a
    patchperpix_amd.minizarr / zarr array, or anything sliceable like one: pred_box() decodes only
    the chunks that intersect the box, straight into a pinned host buffer, and copies that to the
    device (zarr chunk -> pinned host -> HBM; the reference reads block-wise through
    io_hdflike.IoZarr.read as well, :67-120).  Neither the host nor the device ever holds the
    whole prediction.  ``prefetch(box)`` -- called by the tiled assembly with the box it will ask
    for next -- decodes that box into a second pinned buffer on a worker thread while the device
    works on the current tile (chunk decompression runs in C, outside the interpreter lock).
    expit: apply the logistic function to logits (loadAffinities decides that from the value
    range of the whole array, utilVoteInstances.py:249-250; a provider is told)."""

    def __init__(self, arr, device="cuda", expit=False):
        self.arr, self.device, self.expit = arr, device, bool(expit)
        self._pinned = [None, None]
        self._cur = 0
        self._ahead = None            # (box, thread, buffer index, error holder, host buffer, upload slot)
        self._copy_stream = None
        self._upload_events = [None, None]      # per pinned buffer: the last asynchronous copy out of it
        self.boxes_uploaded_ahead = 0
        self.bytes_read = 0
        self.boxes_prefetched = 0

    def _torch_dtype(self):
        import torch
        return {np.dtype(np.float16): torch.float16, np.dtype(np.float32): torch.float32}.get(np.dtype(self.arr.dtype))

    def _shape(self, box):
        z0, z1, y0, y1, x0, x1 = box
        return (int(self.arr.shape[0]), z1 - z0, y1 - y0, x1 - x0)

    def _buffer(self, which, shape):
        import torch
        n, tdt = int(np.prod(shape)), self._torch_dtype()
        buf = self._pinned[which]
        if buf is None or buf.numel() < n or buf.dtype != tdt:
            buf = self._pinned[which] = torch.empty((n,), dtype=tdt, pin_memory=str(self.device).startswith("cuda"))
        return buf[:n].view(shape)

    def _read(self, box, host, err, upload=None, which=None):
        z0, z1, y0, y1, x0, x1 = box
        try:
            if which is not None and self._upload_events[which] is not None:
                self._upload_events[which].synchronize()     # the copy that last read this pinned buffer
                self._upload_events[which] = None
            self.arr.read_into((slice(None), slice(z0, z1), slice(y0, y1), slice(x0, x1)), host.numpy())
            if upload is not None:
                self._upload(host, upload)
                if which is not None:
                    self._upload_events[which] = upload.get("event")
        except BaseException as e:          # reported by the pred_box that consumes the buffer
            err.append(e)

    def _upload(self, host, slot):
        """Worker thread (round 6): the box it has just decoded goes host -> HBM at once, on a copy
        stream of its own, while the device still works on the tile before -- pred_box() then only
        waits for an event.  slot: a dict that receives the device tensor and the event.  Without
        room for a second tile on the device (the allocator says so) the copy stays where it was,
        in pred_box()."""
        import torch
        if not str(self.device).startswith("cuda") or os.environ.get("PPP_ASYNC_H2D", "1") == "0":
            return
        oom = getattr(torch, "OutOfMemoryError", None) or torch.cuda.OutOfMemoryError
        try:
            dev = torch.device(self.device)
            if self._copy_stream is None:
                self._copy_stream = torch.cuda.Stream(device=dev)
            with torch.cuda.device(dev), torch.cuda.stream(self._copy_stream):
                slot["dev"] = host.to(dev, non_blocking=True)
                slot["event"] = self._copy_stream.record_event()
        except oom:
            slot.clear()

    def prefetch(self, box):
        """Start decoding `box` (None: nothing) in the background."""
        import threading
        if box is None or self._torch_dtype() is None or not hasattr(self.arr, "read_into"):
            return
        box = tuple(int(v) for v in box)
        if self._ahead is not None:
            if self._ahead[0] == box:
                return
            self._ahead[1].join()
        which = 1 - self._cur
        host, err, slot = self._buffer(which, self._shape(box)), [], {}
        th = threading.Thread(target=self._read, args=(box, host, err, slot, which), daemon=True)
        th.start()
        self._ahead = (box, th, which, err, host, slot)

    def pred_box(self, box):
        import torch
        box = tuple(int(v) for v in box)
        shape = self._shape(box)
        C = shape[0]
        uploaded = None
        if self._torch_dtype() is None or not hasattr(self.arr, "read_into"):
            z0, z1, y0, y1, x0, x1 = box
            sel = (slice(None), slice(z0, z1), slice(y0, y1), slice(x0, x1))
            host = torch.from_numpy(np.ascontiguousarray(np.asarray(self.arr[sel], dtype=np.float32)))
        else:
            ahead, self._ahead = self._ahead, None
            if ahead is not None:
                ahead[1].join()
            if ahead is not None and ahead[0] == box:
                if ahead[3]:
                    raise ahead[3][0]
                host, self._cur = ahead[4], ahead[2]
                self.boxes_prefetched += 1
                if ahead[5].get("dev") is not None:
                    # already on its way to the device: the consumer's stream waits for the copy
                    uploaded = ahead[5]["dev"]
                    torch.cuda.current_stream().wait_event(ahead[5]["event"])
                    uploaded.record_stream(torch.cuda.current_stream())
                    self.boxes_uploaded_ahead += 1
            else:
                host = self._buffer(self._cur, shape)
                err = []
                self._read(box, host, err, which=self._cur)
                if err:
                    raise err[0]
        self.bytes_read += host.numel() * host.element_size()
        if uploaded is not None:
            pred = uploaded
        else:
            pred = host.to(self.device, non_blocking=True)
            if str(self.device).startswith("cuda"):
                torch.cuda.current_stream().synchronize()     # the pinned buffer is reused two boxes on
            elif pred.data_ptr() == host.data_ptr():
                pred = pred.clone()
        if self.expit:
            # loadAffinities: scipy.special.expit of the float16 array (evaluated in float64),
            # later narrowed to float32 -- the same two roundings here, a few channels at a time
            out = torch.empty(pred.shape, dtype=torch.float32, device=pred.device)
            step = max(1, (1 << 26) // max(1, int(np.prod(shape[1:]))))
            for c0 in range(0, C, step):
                out[c0:c0 + step] = torch.sigmoid(pred[c0:c0 + step].double()).float()
            pred = out
        return pred.contiguous()


# ------------------------------------------------------------------------------------------
# device operations (the C ABI); tests substitute an oracle-backed object with the same API
# ------------------------------------------------------------------------------------------
class DeviceOps:
    def __init__(self, device="cuda"):
        import torch
        self.torch = torch
        self.device = torch.device(device)

    def consensus(self, pred, ov, P):
        return backend.consensus(pred, ov if P.use_overlap else None, P)

    def pred_check(self, pred, P):
        return backend.pred_check(pred, P)

    def rank_patches(self, pred, cons, ov, P, score_box, out=None):
        return backend.rank_patches(pred, cons, ov if P.use_overlap else None, P,
                                    score_box=score_box, out=out)

    def rank_on_voxel_major(self, P):
        return backend.rank_vm_available(P)

    def to_voxel_major(self, cons, P):
        return backend.cons_to_voxel_major(cons, P)

    def consensus_voxel_major(self, pred, ov, P, out=None):
        # (rows only read by the ranking and patch-graph kernels of patches inside the volume:
        # entries beyond the box need no zeroing, ppp_consensus_rows)
        open_rows = os.environ.get("PPP_VM_OPEN", "1") != "0"
        if out is not None and os.environ.get("PPP_VM_POISON") == "1":
            # test switch: the entries open rows leave unwritten must never reach a result --
            # make them NaN (the rank / patch-graph kernels multiply, they do not select)
            out.fill_(float("nan"))
        return backend.consensus_voxel_major(pred, ov if P.use_overlap else None, P, out=out,
                                             open_rows=open_rows)

    def voxel_major_pool(self, P, n_voxels):
        """One flat buffer for the voxel-major consensus of every tile (both passes) when S1
        writes that layout directly; None otherwise."""
        if not backend.direct_voxel_major(P):
            return None
        W = (2 * P.pz - 1) * (2 * P.py - 1) * (2 * P.px - 1)
        return backend._big_empty((int(n_voxels) * W,), self.device)

    # -- consensus cache: COMPACT planes over a rank's whole block, every base voxel computed once
    def cons_cache_alloc(self, P):
        """None when S1 cannot fill a cache for these parameters (packed kernel only)"""
        if not backend.direct_voxel_major(P):
            return None
        Pc = P.copy()
        Pc.cons_layout = backend.CONS_COMPACT
        n = int(backend.lib().ppp_cons_elems(ctypes.byref(Pc)))
        return backend._big_empty((n,), self.device)

    def cons_cache_fill(self, pred, ov, P, part, cache):
        Pc = P.copy()
        Pc.cons_layout = backend.CONS_COMPACT
        backend.consensus_part(pred, ov if P.use_overlap else None, Pc, part, cache)

    def cons_from_cache(self, cache, cache_box, P, out=None):
        """the layout ranking and patch graph read (voxel-major rows of P.cons_box)"""
        if out is not None and os.environ.get("PPP_VM_POISON") == "1":
            out.fill_(float("nan"))
        return backend.cons_planes_to_rows(cache, cache_box, P, out=out)

    # -- rows in a ring of z-slices (ppp_params.ring_z): the new slices of a tile written into the
    #    buffer the previous tile of the column left its upper rows in
    def ring_fill(self, pred, ov, P, part, pool):
        if os.environ.get("PPP_VM_POISON") == "1" and getattr(self, "_ring_poison", None) is not pool:
            pool.fill_(float("nan"))               # (test switch: once per buffer, not per tile)
            self._ring_poison = pool
        backend.consensus_part(pred, ov if P.use_overlap else None, P, part, pool)

    def patch_bits(self, pred, centres, thresh, P, scratch=None):
        return backend.patch_bits(pred, centres, thresh, P, scratch=scratch)

    def patch_pairs(self, sorted_zyx, P, max_ps_dist, include_single):
        return backend.device_patch_pairs(sorted_zyx, P, max_ps_dist=max_ps_dist,
                                          include_single=include_single)

    def patch_graph(self, pred, cons, rows, P):
        return backend.patch_graph_auto(pred, cons, rows, P)

    # streaming pair rows / labels: the rows of a tile exist only while the tile is worked on
    def pair_counts(self, sorted_zyx, subset, P, max_ps_dist):
        return backend.pair_counts_subset(sorted_zyx, subset, P, max_ps_dist=max_ps_dist)

    def pairs_subset(self, sorted_zyx, subset, counts, goffsets, n_rows_total, P, max_ps_dist,
                     include_single):
        return backend.pairs_subset(sorted_zyx, subset, counts, goffsets, n_rows_total, P,
                                    max_ps_dist=max_ps_dist, include_single=include_single)

    def label_state(self, nodes, P):
        return backend.LabelState(nodes, P)

    def cover_shard(self, mask_local, lin_local, rank_id, bits, P_local, global_z):
        return backend.CoverShard(mask_local, lin_local, rank_id, bits, P_local, global_z)

    def rank_order(self, score_dev, foreground, ps):
        """(lin int64, scores float32) of the ranked list, device tensors.  foreground: uint8
        0 / 1 device tensor (Z, Y, X)."""
        return backend.rank_order_device(score_dev, foreground, ps, to_host=False)

    def greedy_cover(self, mask_to_cover, bits, lin, scores, never, pix_ths, radslice, P, kw,
                     bits_first_voxel=None):
        """Greedy cover of the global mask (uint8 0 / 1 device tensor, not modified; replicated:
        every rank runs the same rounds on its own device).  bits: rows in list order, or one row
        per voxel from linear voxel `bits_first_voxel` on.  Returns a bool tensor over the ranked
        list."""
        from .vote_instances import foreground_cover as fc
        sel, _ = fc.greedy_cover_device(mask_to_cover.clone(), bits, lin, never, pix_ths, radslice, P,
                                        bits_first_voxel=bits_first_voxel)
        return sel

    def thin_cover(self, mask_to_cover, bits, lin, P):
        return backend.thin_cover_device(mask_to_cover, bits, lin, P)

    def thin_shard(self, mask_local, lin_local, index_global, bits, P_local, global_z):
        return backend.ThinShard(mask_local, lin_local, index_global, bits, P_local, global_z)

    times_rank_tiles = True       # (the ranking kernel's tile shape is chosen by timing: _Assembly.scores_pass)

    def mws_labels(self, rows, aff, nodes, P):
        return backend.mws_labels_device(rows, aff, nodes, P)

    def label_components(self, rows, aff, nodes, P):
        return backend.label_components(rows, aff, nodes, P)

    def paint(self, pred, nodes, labels, inst, P):
        return backend.paint_instances(pred, nodes, labels, inst, P)


# ------------------------------------------------------------------------------------------
# the greedy cover, sharded by z over the ranks
# ------------------------------------------------------------------------------------------
COVER_BATCH = 8          # rounds between two "is anybody still undecided" all-reduces
INT32_MAX = 0x7FFFFFFF
INT64_MAX = 0x7FFFFFFFFFFFFFFF


def sharded_cover(ops, comm, shape, ps, my_range, ranges, mask_to_cover, lin_t, never, pix_ths,
                  radslice, own_bits, make_local_params):
    """The sharded cover on a REPLICATED ranked list (lin_t int64 [n] / never bool [n]: the global
    ranked list on every rank; mask_to_cover: uint8 0 / 1 device tensor of the whole volume):
    every rank decides the patches centred in its own z-range (sharded_cover_own) and the
    decisions are gathered.  own_bits(idx) -> patch bits of the listed ranked patches (all own).
    Returns selected bool [n] (device tensor), identical on every rank."""
    import torch
    dev = ops.device
    Z, Y, X = [int(v) for v in shape]
    h = int(ps[0]) - 1
    z0, z1 = my_range
    a, b = max(0, z0 - h), min(Z, z1 + h)
    n = int(lin_t.numel())
    cz = torch.div(lin_t, Y * X, rounding_mode="floor")
    own_idx = torch.nonzero((cz >= z0) & (cz < z1)).reshape(-1)          # ascending = rank order
    del cz
    remaining = int(torch.count_nonzero(mask_to_cover[tuple(radslice)]).item())
    sel_own = sharded_cover_own(ops, comm, shape, ps, my_range, ranges, mask_to_cover[a:b].clone(), a,
                                remaining, lin_t[own_idx], own_idx.to(torch.int32), never[own_idx],
                                own_bits(own_idx), pix_ths, make_local_params)
    selected = torch.zeros(n, dtype=torch.bool, device=dev)
    for part in gather_lists(comm, own_idx[sel_own]):
        selected[part] = True
    return selected


def sharded_cover_own(ops, comm, shape, ps, my_range, ranges, mask_ab, a, remaining, lin_own,
                      rank_id, never_own, bits_own, pix_ths, make_local_params):
    """The priority-parallel greedy cover (csrc/ppp_cover.hip) with the volume split by z: every
    rank runs the rounds on its own slices + a halo of pz-1 slices and decides its OWN patches;
    twice per round the 2(pz-1) slices around every slab boundary are made consistent (rank
    volume after the count step; mask and dirty marks after the select step) -- point to point
    between the two neighbours, or by a MIN all-reduce when slabs are thinner than the zones.
    Same result as the sequential cover.

    my_range (z0, z1), ranges: every rank's (z0, z1) in rank order (contiguous, ascending);
    mask_ab   uint8 0 / 1 device tensor, the mask on the global slices [a, b) = own +- (pz - 1)
              (clipped; cleared in place by the rounds);  remaining: interior mask voxels of the
              WHOLE volume (the loop's stop rule);
    lin_own   int64 [m] GLOBAL linear indices of the own ranked patches in rank order,
    rank_id   int32 [m] their positions in the global ranked list, never_own bool [m],
    bits_own  int32 [m, words] their patch bits;
    make_local_params(a, b) -> ppp_params for the local buffer of global slices [a, b).
    Returns selected bool [m] (device tensor)."""
    import torch
    dev = ops.device
    Z, Y, X = [int(v) for v in shape]
    h = int(ps[0]) - 1
    z0, z1 = my_range
    b = a + int(mask_ab.shape[0])
    plane = Y * X
    m_own = int(lin_own.numel())
    shard = ops.cover_shard(mask_ab, (lin_own - a * plane).contiguous(), rank_id.contiguous(), bits_own,
                            make_local_params(a, b), Z)
    # zones around the internal slab boundaries (global slices), and which of them touch me
    bounds = [int(r[1]) for r in ranges[:-1]]
    zones = [(max(0, zb - h), min(Z, zb + h)) for zb in bounds]
    mine = [i for i, zb in enumerate(bounds) if zb == z0 or zb == z1]
    zlen = 2 * h * plane
    rank_buf = torch.empty((max(len(zones), 1), zlen), dtype=torch.int32, device=dev)
    mask_buf = torch.empty((max(len(zones), 1), 2, zlen), dtype=torch.uint8, device=dev)
    own_loc = (z0 - a, z1 - a)

    # a boundary zone concerns the two ranks next to it -- unless slabs are thinner than the zones
    # (then a zone reaches a third rank and everybody takes part in an all-reduce)
    pairwise = hasattr(comm, "neighbour_min") and os.environ.get("PPP_COVER_P2P", "1") != "0" and \
        all(r[1] - r[0] >= 2 * h for r in ranges)

    def zone_io(imp, i, with_rank):
        lo_z, hi_z = zones[i][0] - a, zones[i][1] - a
        if with_rank:
            shard.zone(imp, lo_z, hi_z, own_loc, rank=rank_buf[i])
        else:
            shard.zone(imp, lo_z, hi_z, own_loc, mask=mask_buf[i, 0], clean=mask_buf[i, 1])

    def exchange(with_rank):
        if not zones:
            return
        buf = rank_buf if with_rank else mask_buf
        if pairwise:
            for i in mine:
                zone_io(False, i, with_rank)
            comm.neighbour_min([(comm.rank + 1 if bounds[i] == z1 else comm.rank - 1, buf[i]) for i in mine])
        else:
            buf.fill_(INT32_MAX if with_rank else 1)
            for i in mine:
                zone_io(False, i, with_rank)
            comm.all_reduce_min(buf)
        for i in mine:
            zone_io(True, i, with_rank)

    selected = torch.zeros(m_own, dtype=torch.bool, device=dev)
    total_rounds = 0
    for pix_th in pix_ths:
        if remaining <= 0:
            break
        # every pass restarts at rank 0 (the reference passes rpidx by value)
        shard.open(torch.where(selected, 1, torch.where(never_own, 2, 0)).to(torch.int32))
        alive = True
        while alive:
            for _ in range(COVER_BATCH):
                shard.step(shard.COUNT, pix_th)
                exchange(True)
                shard.step(shard.FILTER)
                shard.step(shard.SELECT, pix_th)
                exchange(False)
            total_rounds += COVER_BATCH
            flag = torch.tensor([1 if shard.alive() else 0], dtype=torch.int32, device=dev)
            alive = int(comm.all_reduce_max(flag).item()) > 0
        shard.close()
        # The loop's stop rule: it ends right after the patch that empties the interior.  The
        # patches selected in this pass, in rank order over ALL ranks, with the interior voxels
        # each cleared: (rank id, cleared) of the own ones are gathered -- a list as long as the
        # selection, not as the ranked list.
        new_own = torch.nonzero((shard.state[:m_own] == 1) & ~selected).reshape(-1) if m_own else \
            torch.zeros((0,), dtype=torch.int64, device=dev)
        mine_rc = torch.stack([rank_id[new_own].to(torch.int64), shard.cleared[:m_own][new_own].to(torch.int64)], 1) \
            if m_own else torch.zeros((0, 2), dtype=torch.int64, device=dev)
        allrc = torch.cat(gather_lists(comm, mine_rc), 0)
        cut = None
        if allrc.shape[0]:
            order = torch.argsort(allrc[:, 0])
            left = remaining - torch.cumsum(allrc[order, 1], 0)
            done = torch.nonzero(left <= 0).reshape(-1)
            if done.numel():
                cut = int(allrc[order[int(done[0].item())], 0].item())     # last rank id the loop reaches
                remaining = 0
            else:
                remaining = int(left[-1].item())
        if m_own:
            keep = new_own if cut is None else new_own[rank_id[new_own].to(torch.int64) <= cut]
            selected[keep] = True
        if remaining < 1:
            break
    backend.note("cover_rounds", total_rounds)
    backend.note("cover_sharded", comm.world)
    backend.note("cover_p2p", 1 if pairwise else 0)
    return selected


THIN_MAXC = 1 << 20          # csrc/ppp_cover.hip: key = (THIN_MAXC - count) << 32 | index


def sharded_thin_own(ops, comm, shape, ps, my_range, ranges, mask_ab, a, interior_total, lin_own, index_own,
                     bits_own, make_local_params):
    """The set-cover thinning (foreground_cover.py:183-256; priority-parallel rounds, csrc/ppp_cover.hip) with
    the volume split by z, after the pattern of sharded_cover_own: every rank runs the rounds on its own
    slices + pz - 1 halo slices and decides its OWN selected patches; twice per round the 2(pz - 1) slices
    around every slab boundary are made consistent -- the 64-bit keys (count, position in the global list)
    after the count step, mask and dirty marks after the select step -- point to point between the two
    neighbours (or by MIN all-reduces when slabs are thinner than the zones).

    mask_ab  uint8 device tensor, the mask on the global slices [a, b) = own +- (pz - 1) (clipped; cleared in
             place); interior_total: interior mask voxels of the WHOLE volume before the thinning (the loop
             stops when they are used up); lin_own int64 [m]: GLOBAL linear indices of the own selected
             patches, index_own int64 [m]: their positions in the global selected list; bits_own [m, words].
    Returns (kept_index int64 [k] -- the positions in the global list of ALL kept patches, ascending, the same
    on every rank --, rounds)."""
    import torch
    dev = ops.device
    Z, Y, X = [int(v) for v in shape]
    h = int(ps[0]) - 1
    z0, z1 = my_range
    b = a + int(mask_ab.shape[0])
    plane = Y * X
    m_own = int(lin_own.numel())
    shard = ops.thin_shard(mask_ab, (lin_own - a * plane).contiguous(), index_own, bits_own, make_local_params(a, b), Z)
    bounds = [int(r[1]) for r in ranges[:-1]]
    zones = [(max(0, zb - h), min(Z, zb + h)) for zb in bounds]
    mine = [i for i, zb in enumerate(bounds) if zb == z0 or zb == z1]
    zlen = 2 * h * plane
    key_buf = torch.empty((max(len(zones), 1), zlen), dtype=torch.int64, device=dev)
    mask_buf = torch.empty((max(len(zones), 1), 2, zlen), dtype=torch.uint8, device=dev)
    own_loc = (z0 - a, z1 - a)
    pairwise = hasattr(comm, "neighbour_min") and os.environ.get("PPP_COVER_P2P", "1") != "0" and \
        all(r[1] - r[0] >= 2 * h for r in ranges)

    def zone_io(imp, i, with_key):
        lo_z, hi_z = zones[i][0] - a, zones[i][1] - a
        if with_key:
            shard.zone(imp, lo_z, hi_z, own_loc, key=key_buf[i])
        else:
            shard.zone(imp, lo_z, hi_z, own_loc, mask=mask_buf[i, 0], clean=mask_buf[i, 1])

    def exchange(with_key):
        if not zones:
            return
        buf = key_buf if with_key else mask_buf
        if pairwise:
            for i in mine:
                zone_io(False, i, with_key)
            comm.neighbour_min([(comm.rank + 1 if bounds[i] == z1 else comm.rank - 1, buf[i]) for i in mine])
        else:
            buf.fill_(INT64_MAX if with_key else 1)
            for i in mine:
                zone_io(False, i, with_key)
            comm.all_reduce_min(buf)
        for i in mine:
            zone_io(True, i, with_key)

    rounds, alive = 0, True
    while alive:
        for _ in range(COVER_BATCH):
            shard.step(shard.COUNT)
            exchange(True)
            shard.step(shard.FILTER)
            shard.step(shard.SELECT)
            exchange(False)
        rounds += COVER_BATCH
        flag = torch.tensor([1 if shard.alive() else 0], dtype=torch.int32, device=dev)
        alive = int(comm.all_reduce_max(flag).item()) > 0
    shard.close()
    # ---- the stop rule over ALL ranks' kept patches: the order the sequential loop picks them in is the
    # order of their keys when they were kept (count descending, index ascending); it stops at the first
    # pick that leaves no interior voxel uncovered
    kept = torch.nonzero(shard.state[:m_own] == 1).reshape(-1) if m_own else torch.zeros((0,), dtype=torch.int64, device=dev)
    key = ((THIN_MAXC - shard.count[:m_own][kept].to(torch.int64)) << 32) | index_own[kept].to(torch.int64)
    mine_kc = torch.stack([key, shard.cleared[:m_own][kept].to(torch.int64)], 1) if m_own else \
        torch.zeros((0, 2), dtype=torch.int64, device=dev)
    allkc = torch.cat(gather_lists(comm, mine_kc), 0)
    keep_idx = torch.zeros((0,), dtype=torch.int64, device=dev)
    remaining = int(interior_total)
    if allkc.shape[0]:
        order = torch.argsort(allkc[:, 0])
        left = remaining - torch.cumsum(allkc[order, 1], 0)
        done = torch.nonzero(left <= 0).reshape(-1)
        n_keep = int(done[0].item()) + 1 if done.numel() else int(order.numel())
        remaining = int(left[n_keep - 1].item())
        keep_idx = allkc[order[:n_keep], 0] & 0xFFFFFFFF
    if remaining > 0:
        # every count is 0 with voxels left: np.argmax picks patch 0 once more (foreground_cover.py:210-216)
        keep_idx = torch.cat([keep_idx, torch.zeros((1,), dtype=torch.int64, device=dev)])
    backend.note("thin_rounds", rounds)
    backend.note("thin_sharded", comm.world)
    return torch.unique(keep_idx), rounds


# ------------------------------------------------------------------------------------------
# the slab pipeline
# ------------------------------------------------------------------------------------------
def _plain(a):
    """bool arrays as uint8 (torch.from_numpy has no bool view of NumPy's bool on every version)"""
    a = np.asarray(a)
    return a.view(np.uint8) if a.dtype == np.bool_ and a.flags.c_contiguous else \
        (a.astype(np.uint8) if a.dtype == np.bool_ else a)


def _field_u8(a, dev, torch):
    """A (Z, Y, X) field (NumPy array or tensor, any integer / bool type) as a uint8 0 / 1
    tensor on `dev`."""
    t = a.to(dev) if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(_plain(a))).to(dev)
    if t.dtype == torch.uint8 and not torch.is_tensor(a) and np.asarray(a).dtype == np.bool_:
        return t.contiguous()
    return (t != 0).to(torch.uint8).contiguous()


class _Frame:
    """A box of the volume with the prediction (and the overlap mask) of exactly that box on the
    device: what one kernel launch sees.  origin = global coordinate of local voxel (0, 0, 0)."""
    __slots__ = ("pred", "ov", "origin", "shape", "clean")

    def __init__(self, pred, ov, origin, shape):
        self.pred, self.ov, self.origin, self.shape = pred, ov, tuple(origin), tuple(shape)
        self.clean = 0          # ppp_params.pred_clean of this frame's tensor: decided at its first use


def assemble(pred_local, lo, shape, foreground, mask_to_cover, numinst, patchshape, my_slabs,
             comm=None, ops=None, **kw):
    """vote_instances on a z-slab decomposition.

    pred_local  the prediction this rank can see: a (C, hi-lo, Y, X) tensor on ops.device holding
                global z in [lo, hi) -- or a PROVIDER, an object with ``pred_box((z0, z1, y0, y1,
                x0, x1)) -> (C, bz, by, bx) tensor`` that produces (reads, generates) the
                prediction of a box on demand, so that a rank holds one tile + halo at a time
                (BASELINE config [3]: a rank's slab + halo of the 1024^3 / 9^3 volume is 263 GB);
    shape       global (Z, Y, X);
    foreground / mask_to_cover / numinst: the per-voxel fields, NumPy arrays or tensors, either
                GLOBAL (Z, Y, X) or LOCAL: the slices [lo, hi) only (hi = lo + their length; a
                rank needs its own slabs +- halo(patchshape), clipped).  With local fields -- or
                ``_sharded_global=True`` -- nothing a rank holds or computes is as large as the
                whole volume: the ranked list, the cover and the sort are sharded by z (stage B2);
    my_slabs    [(z0, z1), ...] owned by this rank (contiguous, inside [lo, hi) minus halo).
    Returns (instances (Z,Y,X) -- complete on every rank; uint16, or uint32 with
    ``_instances_dtype`` --, foreground uint8); with ``_gather_result=False`` the own z-range of
    both only; or (pairs, aff) with return_intermediates, with the reference's early-outs.
    """
    return _Assembly(pred_local, lo, shape, foreground, mask_to_cover, numinst, patchshape, my_slabs, comm, ops, kw).run()


class _Assembly:
    """The state of one tiling.assemble() call; every stage is a method (round 6: up to round 5 this
    was ONE function of 920 lines whose stages were nested closures over forty shared locals).  An
    attribute is what more than one stage -- or a helper closure -- reads; a stage returns None to go
    on, or the call's result (the reference's early-outs, the intermediates, the instance map)."""

    def __init__(self, pred_local, lo, shape, foreground, mask_to_cover, numinst, patchshape, my_slabs, comm, ops, kw):
        self.pred_local = pred_local
        self.lo = lo
        self.shape = shape
        self.foreground = foreground
        self.mask_to_cover = mask_to_cover
        self.numinst = numinst
        self.patchshape = patchshape
        self.my_slabs = my_slabs
        self.comm = comm
        self.ops = ops
        self.kw = kw

    STAGES = ("setup", "frames", "rows_plan", "scores_pass", "select_patches", "thin_cover", "pair_affinities",
              "label_and_paint")

    def run(self):
        try:
            for name in self.STAGES:
                result = getattr(self, name)()
                if result is not None:
                    return result
            raise AssertionError("the last stage returns the result")
        finally:
            # the helper closures hold `self` and `self` holds them: without this the tensors among the
            # attributes (fields, node lists, an exchanged slab) would wait for the cycle collector
            self.__dict__.clear()

    def setup(self):
        """arguments checked, the per-voxel fields on the device, the halo exchange, the early-outs"""
        import torch
        # flags the slab pipeline does not implement are refused, never ignored (the caller --
        # to_instance_seg -- applies skeletonize_foreground to the mask before it gets here)
        for opt in ("skipConsensus", "skipRanking", "termAfterThinCover", "termAfterPatchGraph",
                    "save_consensus", "graphToInst", "debug", "isbiHack", "pad_with_ps",
                    "one_instance_per_channel", "no_overlap_per_channel", "sparse_labels"):
            if self.kw.get(opt, False):
                raise NotImplementedError("%s is not supported by the tiled / multi-rank assembly" % opt)
        # the two optional branches of the greedy cover (foreground_cover.py:53-85, 141-168) leave marks
        # anywhere in a slice: a sequential walk of the ranked list -- served on ONE rank (below)
        self.seq_cover = bool(self.kw.get("mark_close_neighboorhood", False) or self.kw.get("select_patches_overlap_neighborhood", False))
        if self.kw.get("aff_graph") is not None:
            raise NotImplementedError("aff_graph input is not supported by the tiled assembly")
        if self.kw.get("consensus_interleaved_cnt", True):
            # consensus_array.py:131-133, where the reference launches its consensus kernel
            assert self.kw.get("consensus_norm_aff", True), "consensus aff not normalized so no computation required"
        if self.kw.get("max_total_patch_distance_in_ps_multiples", 2) > 2:
            raise NotImplementedError("the slab halo is sized for "
                                      "max_total_patch_distance_in_ps_multiples <= 2")
        if not self.my_slabs:
            raise ValueError("this rank owns no z-slab (more ranks than slabs)")
        self.comm = self.comm or LocalComm()
        if self.seq_cover and self.comm.world > 1:
            raise NotImplementedError("mark_close_neighboorhood / select_patches_overlap_neighborhood walk the ranked "
                                      "list sequentially: one rank only")
        self.ops = self.ops or DeviceOps()
        self.dev = self.ops.device
        self.Z, self.Y, self.X = [int(s) for s in self.shape]
        self.dims = (self.Z, self.Y, self.X)
        self.plane = self.Y * self.X
        self.ps = [int(p) for p in self.patchshape]
        self.rz = self.ps[0] // 2
        self.rad = np.array([p // 2 for p in self.ps])
        self.H = halo(self.ps)
        self.provider = not torch.is_tensor(self.pred_local)
        # ---- the per-voxel fields: on the device once (or already device tensors), in the "field
        # frame" [flo, fhi) -- the whole volume, or the slices this rank was given.  Every later use
        # (early-outs, overlap mask of the kernels, ranking, cover, thinning) is a device operation:
        # host NumPy passes over a 512^3 volume cost 0.1-0.3 s each, and the reference's
        # `1 * (numinst > 1)` is an int64 temporary of 8 bytes per voxel.
        self.Zf = int(self.foreground.shape[0])
        self.local_fields = self.Zf != self.Z
        # "some rank holds local fields": what every COLLECTIVE decision below asks (frames() settles it over
        # the ranks) -- on a short volume a middle rank's slices + halo can be the whole volume while its
        # neighbours' are not, and the ranks must still take the same path
        self.any_local = self.local_fields
        self.flo = self.lo if self.local_fields else 0
        fhi = self.flo + self.Zf
        if self.provider:
            self.hi = fhi if self.local_fields else self.Z
            lo_p = self.lo if self.local_fields else 0
        else:
            self.hi = self.lo + int(self.pred_local.shape[1])
            lo_p = self.lo
            if self.local_fields and self.Zf != self.hi - self.lo:
                raise ValueError("local fields must cover the slices of pred_local")
        for f in (self.mask_to_cover, self.numinst):
            if tuple(int(v) for v in f.shape) != (self.Zf, self.Y, self.X):
                raise ValueError("foreground / mask_to_cover / numinst differ in shape")
        self.oz0, self.oz1 = int(self.my_slabs[0][0]), int(self.my_slabs[-1][1])       # hull of the own slabs
        self.contiguous = all(self.my_slabs[i][1] == self.my_slabs[i + 1][0] for i in range(len(self.my_slabs) - 1))
        self.need_lo, self.need_hi = max(0, self.oz0 - self.H), min(self.Z, self.oz1 + self.H)
        if self.comm.world > 1 and self.contiguous and self.kw.get("_exchange_halo", True):
            # A RESIDENT slab (and local fields) that lack the halo -- the U-Net's output as it stands on
            # each GPU -- get it from the ranks that own it (exchange_halo: point to point over RCCL / xGMI).
            # Every rank takes part in the decision: one MAX all-reduce of "I miss slices".
            # `_refresh_halo`: pred_local is a halo-sized buffer whose OWN slices are current (the producer
            # wrote them) and whose halo slices are stale -- fetched again, in place
            refresh = bool(self.kw.get("_refresh_halo", False)) and not self.provider
            miss_p = (not self.provider) and (lo_p > self.need_lo or self.hi < self.need_hi or refresh)
            miss_f = self.local_fields and (self.flo > self.need_lo or fhi < self.need_hi)
            flag = torch.tensor([int(miss_p), int(miss_f)], dtype=torch.int32, device=self.ops.device)
            any_p, any_f = [int(v) for v in self.comm.all_reduce_max(flag).cpu()]
            if any_p and self.provider:
                raise ValueError("halo exchange: every rank must hold a resident prediction slab")
            with backend.host_timer("halo_exchange"):
                if any_p:
                    own = self.pred_local.narrow(1, self.oz0 - self.lo, self.oz1 - self.oz0)
                    in_place = refresh and self.lo <= self.need_lo and self.hi >= self.need_hi
                    self.pred_local = exchange_halo(own, (self.oz0, self.oz1), (self.need_lo, self.need_hi), self.comm, z_axis=1,
                                               out=self.pred_local.narrow(1, self.need_lo - self.lo, self.need_hi - self.need_lo) if in_place else None)
                    self.lo = lo_p = self.need_lo
                    self.hi = self.need_hi
                if any_f:
                    def with_halo(f):
                        t = f if torch.is_tensor(f) else torch.from_numpy(np.ascontiguousarray(_plain(f)))
                        t = t.to(self.ops.device).narrow(0, self.oz0 - self.flo, self.oz1 - self.oz0)
                        if t.dtype == torch.bool:
                            t = t.to(torch.uint8)
                        return exchange_halo(t, (self.oz0, self.oz1), (self.need_lo, self.need_hi), self.comm, z_axis=0)
                    self.foreground, self.mask_to_cover, self.numinst = with_halo(self.foreground), with_halo(self.mask_to_cover), with_halo(self.numinst)
                    self.flo, fhi = self.need_lo, self.need_hi
                    self.Zf = fhi - self.flo
            if self.local_fields and not self.provider and (self.flo, fhi) != (self.lo, self.hi):
                raise ValueError("local fields must cover the slices of pred_local")
        if self.flo > self.need_lo or fhi < self.need_hi or lo_p > self.need_lo or self.hi < self.need_hi:
            raise ValueError("this rank's slabs [%d, %d) need the slices [%d, %d)" % (self.oz0, self.oz1, self.need_lo, self.need_hi))
        self.fg_d = _field_u8(self.foreground, self.dev, torch)
        ni_d = self.numinst.to(self.dev) if torch.is_tensor(self.numinst) else \
            torch.from_numpy(np.ascontiguousarray(_plain(self.numinst))).to(self.dev)
        self.ov_d = (ni_d > 1).to(torch.uint8)
        del ni_d
        self.mask_d = _field_u8(self.mask_to_cover, self.dev, torch)
        self.want_inter = self.kw.get("return_intermediates", False)
        self.gather_result = self.kw.get("_gather_result", True)
        self.id_dtype = np.dtype(self.kw.get("_instances_dtype") or np.uint16)
        if self.id_dtype not in (np.dtype(np.uint16), np.dtype(np.uint32)):
            raise ValueError("_instances_dtype must be uint16 or uint32")

        def own_sum(t):
            """the same scalar on every rank: sum over ranks of a per-rank count"""
            v = torch.tensor([int(t)], dtype=torch.int64, device=self.dev)
            return int(self.comm.all_reduce_sum(v).item()) if self.comm.world > 1 else int(t)
        self.own_sum = own_sum

        def own_z(z0, z1):
            """field-frame slice of the global slices [z0, z1)"""
            return slice(z0 - self.flo, z1 - self.flo)
        self.own_z = own_z

        def interior_count(field):
            """voxels set in `field` (field frame) inside the interior of the WHOLE volume: counted
            on the own slabs, summed over the ranks"""
            n = 0
            for (z0, z1) in self.my_slabs:
                a, b = max(z0, self.rz), min(z1, self.Z - self.rz)
                if a < b:
                    n += int(torch.count_nonzero(field[self.own_z(a, b), self.rad[1]:self.Y - self.rad[1], self.rad[2]:self.X - self.rad[2]]).item())
            return self.own_sum(n)
        self.interior_count = interior_count

        self.any_overlap = self.own_sum(int(self.ov_d[self.own_z(self.need_lo, self.need_hi)].any().item())) > 0
        if self.any_overlap:
            self.mask_d &= 1 - self.ov_d
            if not torch.is_tensor(self.mask_to_cover):
                # vote_instances.py:226 clears the caller's array as well
                self.mask_to_cover[self.ov_d.cpu().numpy().astype(bool)] = 0

        def fg_out(z0=None, z1=None):
            sl = slice(None) if z0 is None else self.own_z(z0, z1)
            return self.fg_d[sl].cpu().numpy() if torch.is_tensor(self.foreground) else \
                np.asarray(self.foreground[sl]).astype(np.uint8)
        self.fg_out = fg_out

        def full_fg():
            if not self.any_local:
                return self.fg_out()
            g = torch.zeros(self.shape, dtype=torch.uint8, device=self.dev)
            g[self.oz0:self.oz1] = self.fg_d[self.own_z(self.oz0, self.oz1)]
            if self.rank_ranges is not None:
                self.comm.all_gather_slabs(g, self.rank_ranges)
            else:
                self.comm.all_reduce_sum(g)
            return g.cpu().numpy()
        self.full_fg = full_fg

        def early():
            if self.want_inter:
                return None, None
            if not self.gather_result:
                return np.zeros((self.oz1 - self.oz0, self.Y, self.X), dtype=self.id_dtype), self.fg_out(self.oz0, self.oz1)
            return np.zeros(self.shape, dtype=self.id_dtype), self.full_fg()
        self.early = early


    def frames(self):
        """tiles, ranks' ranges, frames (what a kernel launch sees) and their parameters"""
        import torch
        self.flags = {k: v for k, v in self.kw.items()
                 if k not in ("cons_box", "cons_layout", "origin", "_yx_tiles", "_instances_dtype")}
        self.ny_t, self.nx_t = self.kw.get("_yx_tiles") or (1, 1)
        # tiles of patch centres (z0, z1, y0, y1, x0, x1): the rank's z-slabs, each cut in y / x
        self.my_tiles = [(z0, z1) + t for (z0, z1) in self.my_slabs for t in plan_yx(self.Y, self.X, self.ny_t, self.nx_t)]
        # every rank's contiguous z-range (None when a rank's slabs are not contiguous): the owned
        # parts of the score / instance volumes are exchanged by ONE all-gather of slabs
        self.rank_ranges = None
        if self.comm.world > 1:
            mine_r = torch.tensor([self.oz0, self.oz1 if self.contiguous else -1, int(self.local_fields)], dtype=torch.int64, device=self.dev)
            rr = [tuple(int(v) for v in r) for r in self.comm.all_gather(mine_r).cpu().numpy()]
            self.any_local = any(r[2] for r in rr)
            rr = [r[:2] for r in rr]
            if all(r[1] >= 0 for r in rr) and all(rr[i][1] == rr[i + 1][0] for i in range(len(rr) - 1)):
                self.rank_ranges = rr
        else:
            self.rank_ranges = [(self.oz0, self.oz1)] if self.contiguous else None
        self.sharded = bool(self.kw.get("_sharded_global", self.any_local))
        if self.sharded and (self.rank_ranges is None or not hasattr(self.ops, "cover_shard")):
            raise ValueError("the sharded global stage needs one contiguous z-range per rank")
        if self.any_local and not self.sharded:
            raise ValueError("local fields need the sharded global stage")

        if self.interior_count(self.mask_d) == 0 or self.interior_count(self.fg_d) == 0:
            return self.early()

        # ---- frames: what a kernel launch sees.  Resident prediction: ONE frame, the rank's whole
        # local block.  Provider: a frame per tile and pass, the tile grown by what the pass reads.
        def grow(t, g):
            g = [int(v) for v in (g if np.ndim(g) else (g, g, g))]
            return tuple(v for a in range(3) for v in (max(0, t[2 * a] - g[a]), min(self.dims[a], t[2 * a + 1] + g[a])))
        self.grow = grow

        def field_box(field, box):
            z0, z1, y0, y1, x0, x1 = box
            return field[self.own_z(z0, z1), y0:y1, x0:x1].contiguous()
        self.field_box = field_box

        self.whole = None
        if not self.provider:
            self.whole = _Frame(self.pred_local, self.field_box(self.ov_d, (self.lo, self.hi, 0, self.Y, 0, self.X)), (self.lo, 0, 0), (self.hi - self.lo, self.Y, self.X))

        def frame_for(box, next_box=None):
            """next_box: the box that will be asked for after this one (a provider that can work
            ahead -- ZarrProvider -- decodes it while this tile is on the device)"""
            if self.whole is not None:
                return self.whole
            z0, z1, y0, y1, x0, x1 = box
            with backend.host_timer("provider"):
                pred_t = self.pred_local.pred_box(box)
                if next_box is not None and hasattr(self.pred_local, "prefetch"):
                    self.pred_local.prefetch(next_box)
            backend.note_add("provider_voxels", (z1 - z0) * (y1 - y0) * (x1 - x0))
            return _Frame(pred_t, self.field_box(self.ov_d, box), (z0, y0, x0), (z1 - z0, y1 - y0, x1 - x0))
        self.frame_for = frame_for

        def params(fr, box=None):
            """box: global (z0, z1, y0, y1, x0, x1) of consensus base voxels, or None."""
            o = fr.origin
            if box is not None:
                box = (box[0] - o[0], box[2] - o[1], box[4] - o[2], box[1] - o[0], box[3] - o[1], box[5] - o[2])
            P = backend.make_params(fr.shape, self.ps, cons_box=box, origin=o, **self.flags)
            if fr.clean == 0 and fr.pred is not None and hasattr(self.ops, "pred_check"):
                # ONE streaming pass per frame (the resident block: once per call; a provider's box: once
                # per box) tells every S1 launch on it whether the short classification applies
                fr.clean = self.ops.pred_check(fr.pred, P)
            P.pred_clean = fr.clean
            return P
        self.params = params

        def to_local(coords_t, fr, cols=1):
            """global (z, y, x) [x cols] -> frame coordinates"""
            sh = torch.tensor(list(fr.origin) * cols, dtype=coords_t.dtype, device=self.dev)
            return (coords_t - sh).contiguous()
        self.to_local = to_local

        self.Pg = backend.make_params(self.shape, self.ps, **self.flags)          # global geometry (pairs, labels)
        self.keep_cons = len(self.my_tiles) == 1 and self.kw.get("_keep_cons", True)
        self.kept = {}

        def bases_for_scores(t):
            """consensus bases the scores of the centres in tile t read: t grown by the radius"""
            return self.grow(t, self.rad)
        self.bases_for_scores = bases_for_scores

        def bases_for_pairs(t):
            """... the pairs with patch A in t read: the voxel-major rows S[u][q], u in win(A), are
            built from the stored (positive) offsets, S[u][q < 0] = cons[-q][u + q]; u + q lies up
            to p-1 below u in z and up to p-1 on either side in y / x"""
            b = []
            for a in range(3):
                g = self.ps[a] - 1
                b += [max(0, t[2 * a] - int(self.rad[a]) - g),
                      min(self.dims[a], t[2 * a + 1] + int(self.rad[a]) + (g if a > 0 else 0))]
            return tuple(b)
        self.bases_for_pairs = bases_for_pairs

        self.words = (int(np.prod(self.ps)) + 31) // 32
        self.can_vm = hasattr(self.ops, "consensus_voxel_major")

        def consensus_of(fr, P, pool):
            if self.can_vm and self.ops.rank_on_voxel_major(P):
                # ranking and patch graph both read the voxel-major layout: S1 writes it directly
                # where the library can (else compact planes + one re-layout, planes dropped)
                return self.ops.consensus_voxel_major(fr.pred, fr.ov, P, **({"out": pool} if pool is not None else {}))
            return self.ops.consensus(fr.pred, fr.ov, P), P
        self.consensus_of = consensus_of


    def rows_plan(self):
        """how a tile's voxel-major rows come about: consensus cache, ring of rows, the pooled buffer"""
        import torch
        # ---- consensus cache (`_cons_cache`, decided by the caller's memory plan: plan_tiles).
        # Several tiles and a resident prediction: the COMPACT planes of the rank's whole block (every
        # tile's pairs box) are computed ONCE, tile by tile, into one array; the voxel-major rows a tile
        # needs -- for its scores now, for its pair rows after the global stage -- are cut from it by a
        # transpose.  S1 then runs over 1.0 x the block (+ the rank's z-halo) instead of scores pass
        # (tile + radius) + pairs pass (tile + radius + p - 1): 2.5 x at 512^3 / 9^3.
        self.cache = self.cache_box = None
        if self.kw.get("_cons_cache") and not self.keep_cons and len(self.my_tiles) > 1 and self.whole is not None and self.contiguous \
                and hasattr(self.ops, "cons_cache_alloc") \
                and (not hasattr(self.ops, "rank_on_voxel_major") or self.ops.rank_on_voxel_major(self.params(self.whole))):
            boxes = [self.bases_for_pairs(t) for t in self.my_tiles]
            self.cache_box = tuple(f(b[i] for b in boxes) for i, f in enumerate((min, max) * 3))
            Pc = self.params(self.whole, self.cache_box)
            self.cache = self.ops.cons_cache_alloc(Pc)
            if self.cache is None:
                self.cache_box = None
            else:
                def to_frame(b):
                    o = self.whole.origin
                    return (b[0] - o[0], b[2] - o[1], b[4] - o[2], b[1] - o[0], b[3] - o[1], b[5] - o[2])
                self.to_frame = to_frame
                with backend.host_timer("s1_consensus"):
                    for t in self.my_tiles:
                        # a tile fills its own voxels; tiles on the rim of the rank's block also the rim
                        part = list(t)
                        for a in range(3):
                            lo_a, hi_a = (self.oz0, self.oz1) if a == 0 else (0, self.dims[a])
                            if t[2 * a] == lo_a:
                                part[2 * a] = self.cache_box[2 * a]
                            if t[2 * a + 1] == hi_a:
                                part[2 * a + 1] = self.cache_box[2 * a + 1]
                        self.ops.cons_cache_fill(self.whole.pred, self.whole.ov, Pc, self.to_frame(part), self.cache)
                backend.note("cons_cache_gb", round(self.cache.numel() * 4 / 1e9, 2))

        def rows_from_cache(fr, cbox, pool):
            with backend.host_timer("cons_rows"):
                return self.ops.cons_from_cache(self.cache, self.to_frame(self.cache_box), self.params(fr, cbox), **({"out": pool} if pool is not None else {}))
        self.rows_from_cache = rows_from_cache

        # ---- z-sweep with the rows in a RING (`_ring_z`, decided by the caller's memory plan: plan_ring).
        # The tiles of a (y, x) column are visited bottom-up; the row buffer is a ring of `_ring_z`
        # slices (row of slice z in slot z mod ring).  A tile computes only the base slices its
        # predecessor in the column has not (ppp_consensus_part) -- the rows it shares with it are still
        # in the ring, and the mirrored entries S[u + d][-d] its predecessor's last bases wrote into
        # rows ABOVE their own box wait there ("spill": the box of a launch reaches p - 1 slices past
        # its last base).  So neither pass pays a z-halo: S1 runs over (tile + radius in y / x) per
        # pass instead of (tile + radius) and (tile + radius + p - 1) in all three axes.
        self.ring_z = int(self.kw.get("_ring_z") or 0)
        if self.ring_z and (self.keep_cons or self.cache is not None or self.whole is None or not hasattr(self.ops, "ring_fill")
                       or not self.ops.rank_on_voxel_major(self.params(self.whole))):
            self.ring_z = 0
        if self.ring_z:
            # (a patch shape whose ranking kernel reads plain boxes only -- 3^3 -- sweeps plain tiles)
            Pq = self.params(self.whole)
            Pq.ring_z = self.ring_z
            if not self.ops.rank_on_voxel_major(Pq):
                self.ring_z = 0
        if self.ring_z:
            thick = max(t[1] - t[0] for t in self.my_tiles)
            if self.ring_z < thick + ring_margin(self.ps[0]) or thick < self.ps[0] - 1:
                raise ValueError("_ring_z = %d is too small for tiles of %d slices (or the tiles are thinner than "
                                 "p - 1)" % (self.ring_z, thick))
            # column-major order: all z-tiles of one (y, x) column, bottom-up, then the next column
            self.my_tiles.sort(key=lambda t: (t[2], t[4], t[0]))
            backend.note("ring_z", self.ring_z)
        self.ring_state = {}
        # the scores pass keeps a column's rows on a smaller (y, x) box than the pairs pass the pool is
        # sized for (tile + radius instead of tile + radius + p - 1 per side): its ring is as many slices
        # as the same pool holds of THAT box (44 -> 49 at 512^3 / 9^3 with 256^2 columns)
        self.ring_sc = {"z": self.ring_z}

        def ring_rows(fr, t, pairs_pass, pool):
            """S1 for the new base slices of tile t into the ring; returns (pool, params of the rows the
            consumers of t read: slices [z0 - rad, z1 + rad), the column's y / x box)."""
            ybox = self.bases_for_pairs(t) if pairs_pass else self.bases_for_scores(t)
            r0, r1 = self.bases_for_scores(t)[:2]                          # rows the consumers read
            # (the TILE's y / x range names the column, not only the box of its rows: on a narrow axis two
            # columns grow to the same clipped box, and the second one must start its sweep from the bottom)
            col = (pairs_pass,) + tuple(t[2:]) + tuple(ybox[2:])
            hi = self.ring_state.get("hi") if self.ring_state.get("col") == col else None
            if hi is None or hi < r0:
                # first tile of the column: for the pair rows also the p - 1 source slices below
                part0 = max(self.bases_for_pairs(t)[0], 0) if pairs_pass else r0
            else:
                part0 = hi
            o = fr.origin
            top = min(fr.shape[0] + o[0], self.dims[0], r1 + self.ps[0] - 1)     # spill rows (inside the frame)
            ring_len = self.ring_z if pairs_pass else self.ring_sc["z"]
            if part0 < r1:
                P1 = self.params(fr, (min(part0, r0), top) + tuple(ybox[2:]))
                P1.cons_layout = backend.CONS_VOXEL_MAJOR
                P1.ring_z = ring_len
                part = (part0 - o[0], ybox[2] - o[1], ybox[4] - o[2], r1 - o[0], ybox[3] - o[1], ybox[5] - o[2])
                self.ops.ring_fill(fr.pred, fr.ov, P1, part, pool)
            self.ring_state.update(col=col, hi=r1)
            Pr = self.params(fr, (r0, r1) + tuple(ybox[2:]))
            Pr.cons_layout = backend.CONS_VOXEL_MAJOR
            Pr.ring_z = ring_len
            return pool, Pr
        self.ring_rows = ring_rows

        # ---- stage A: consensus + scores per tile ----------------------------------------------
        # several tiles: ONE consensus buffer, sized for the largest box of either pass, serves all
        # of them (allocated first, while the allocator's address space is still unfragmented)
        self.pool = None
        if not self.keep_cons and self.my_tiles and hasattr(self.ops, "voxel_major_pool"):
            biggest = max(int(np.prod([b[2 * a + 1] - b[2 * a] for a in range(3)]))
                          for b in (self.bases_for_pairs(t) for t in self.my_tiles))
            if self.ring_z:
                biggest = self.ring_z * max((b[3] - b[2]) * (b[5] - b[4]) for b in (self.bases_for_pairs(t) for t in self.my_tiles))
            fr0 = self.whole if self.whole is not None else _Frame(None, None, (0, 0, 0), (2 * self.ps[0], 2 * self.ps[1], 2 * self.ps[2]))
            P0 = self.params(fr0)
            if self.ops.rank_on_voxel_major(P0):
                self.pool = self.ops.voxel_major_pool(P0, biggest)
            if self.ring_z and self.pool is not None and os.environ.get("PPP_RING_SCORES", "1") != "0":
                area_sc = max((b[3] - b[2]) * (b[5] - b[4]) for b in (self.bases_for_scores(t) for t in self.my_tiles))
                self.ring_sc["z"] = max(self.ring_z, int(biggest // area_sc))

    def scores_pass(self):
        """stage A: consensus + scores per tile"""
        import torch
        # scores in the field frame; with a provider also the patch bits of every own voxel (the
        # cover candidates), packed while the tile's prediction exists
        self.score_f = torch.zeros((self.Zf, self.Y, self.X), dtype=torch.float32, device=self.dev)
        self.bits_own = None
        if self.provider:
            self.bits_own = torch.zeros(((self.oz1 - self.oz0) * self.plane, self.words), dtype=torch.int32, device=self.dev)
        def pairs_frame_box(t):
            # S1 reads the prediction 2 rad around its bases; the partner patches B of the rows (up
            # to 2 p away) and their windows: the tile grown by the halo
            return self.grow(t, (self.H, 2 * self.ps[1] + int(self.rad[1]), 2 * self.ps[2] + int(self.rad[2])))
        self.pairs_frame_box = pairs_frame_box

        def scores_frame_box(t):
            # (a kept consensus keeps its frame for the patch-graph stage, which looks at the windows of
            # partners up to 2 p away: the larger box -- it holds the bases of the pair rows + 2 rad)
            return pairs_frame_box(t) if self.keep_cons else self.grow(self.bases_for_scores(t), 2 * self.rad)

        # Ring sweep: ONE ranking launch serves as many consecutive tiles of a column as the ring holds
        # rows for (their slices + the radius on both sides + the p - 1 slices S1 reaches past its last
        # base).  A launch is as long as its slowest workgroup (the tiles of centres differ in valid
        # rows and foreground pixels: 146 ms for one round of 1 024 workgroups, + 64 ms for every further
        # round on a 264 x 264 column, profiles/r05_zl_s2_rounds.txt) -- a launch of 2 048 workgroups
        # costs 0.72 of two launches of 1 024.  PPP_RANK_GROUP=1: one launch per tile.
        rank_group = 1
        tiles = self.my_tiles
        split = int(self.kw.get("_scores_split", os.environ.get("PPP_SCORES_SPLIT", "1")))
        if self.ring_z and split > 1 and self.pool is not None:
            # Round 6 (experiment, PPP_SCORES_SPLIT=2): the scores pass on NARROWER columns than the pairs
            # pass.  A ranking launch gets more efficient with its size (7.7 / 10.3 / 11.8 workgroups per ms
            # at 1 024 / 2 048 / 4 096, profiles/r06_e_*) and the ring holds what the pool holds of a
            # column's box: columns of half the width in y and x hold four times the slices, so one launch
            # ranks several tiles of a column -- for a larger y / x halo in this pass's S1.
            z_ranges = sorted(set((t[0], t[1]) for t in self.my_tiles))
            tiles = [(z0, z1) + c for c in plan_yx(self.Y, self.X, self.ny_t * split, self.nx_t * split) for (z0, z1) in z_ranges]
            W_row = int(np.prod([2 * p - 1 for p in self.ps]))
            area_sc = max((b[3] - b[2]) * (b[5] - b[4]) for b in (self.bases_for_scores(t) for t in tiles))
            self.ring_sc["z"] = max(self.ring_z, int(self.pool.numel() // W_row // area_sc))
            backend.note("scores_columns", self.ny_t * split * self.nx_t * split)
        if self.ring_z:
            thick = max(t[1] - t[0] for t in tiles)
            rank_group = max(1, (self.ring_sc["z"] - 2 * int(self.rad[0]) - (self.ps[0] - 1)) // thick)
            rank_group = max(1, min(rank_group, int(os.environ.get("PPP_RANK_GROUP", rank_group))))
            backend.note("rank_group", rank_group)
            backend.note("ring_z_scores", self.ring_sc["z"])
        pending = []          # tiles of the current column whose rows are in the ring, not ranked yet
        # Which tile of centres a ranking workgroup takes (ppp_params.rank_tile) is decided per CALL by
        # measurement (round 6): 8 x 8 x 16 is 3 % faster where a 2 048-tile launch takes 161 ms, 16 x 8 x 16 is
        # 12 % faster on the boxes where it takes 215 ms (the kernel follows the memory system's speed, the
        # boxes differ; profiles/r06_l_bench_ab_tiles.txt) -- the scores are the same bits either way.  With
        # enough launches ahead the first two are timed, one of each shape, and the faster serves the rest.
        # PPP_RANK_TILE=0..3 fixes the choice (0: the library's rule).
        n_launches = -(-len(tiles) // rank_group)
        tile_env = os.environ.get("PPP_RANK_TILE", "auto")
        rank_tile = int(tile_env) if tile_env.isdigit() else 0
        trials = [1, 3] if (tile_env == "auto" and n_launches >= 6 and getattr(self.ops, "times_rank_tiles", False)) else []
        timed = []            # (shape, start event, stop event, centres)

        for ti, t in enumerate(tiles):
            z0, z1, y0, y1, x0, x1 = t
            cbox = self.bases_for_pairs(t) if self.keep_cons else self.bases_for_scores(t)
            fr = self.frame_for(scores_frame_box(t), scores_frame_box(tiles[ti + 1]) if ti + 1 < len(tiles) else None)
            o = fr.origin
            P = self.params(fr, cbox)
            if self.cache is not None:
                cons, P = self.rows_from_cache(fr, cbox, self.pool)
            elif self.ring_z:
                with backend.host_timer("s1_consensus"):
                    cons, P = self.ring_rows(fr, t, False, self.pool)
                if rank_group > 1:
                    pending.append(t)
                    nxt = tiles[ti + 1] if ti + 1 < len(tiles) else None
                    if len(pending) < rank_group and nxt is not None and nxt[2:] == t[2:] and nxt[0] == t[1]:
                        del cons, fr
                        continue                       # ranked together with the next tile of the column
                    # the rows of every pending tile: [first z0 - rad, last z1 + rad) of the column's box
                    z0 = pending[0][0]
                    rb = self.bases_for_scores((z0, z1) + tuple(t[2:]))
                    P = self.params(fr, rb)
                    P.cons_layout = backend.CONS_VOXEL_MAJOR
                    P.ring_z = self.ring_sc["z"]
                    pending = []
            else:
                with backend.host_timer("s1_consensus"):
                    cons, P = self.consensus_of(fr, P, self.pool)
            with backend.host_timer("s2_rank"):
                same = fr.shape == (self.Zf, self.Y, self.X) and o == (self.flo, 0, 0)
                trial = trials.pop(0) if trials else 0
                P.rank_tile = trial or rank_tile
                if trial:
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[0].record(torch.cuda.current_stream())
                sc = self.ops.rank_patches(fr.pred, cons, fr.ov, P, (z0 - o[0], y0 - o[1], x0 - o[2], z1 - o[0], y1 - o[1], x1 - o[2]),
                                      **({"out": self.score_f} if same and hasattr(self.ops, "voxel_major_pool") else {}))
                if trial:
                    ev[1].record(torch.cuda.current_stream())
                    timed.append((trial, ev[0], ev[1], (z1 - z0) * (y1 - y0) * (x1 - x0)))
                    if not trials:
                        torch.cuda.synchronize()
                        per_centre = {shape: a.elapsed_time(b) / max(1, n) for shape, a, b, n in timed}
                        rank_tile = min(per_centre, key=per_centre.get)
                        backend.note("rank_tile", rank_tile)
                        backend.note("rank_tile_trial_ns_per_centre",
                                     " ".join("%d:%.1f" % (k, 1e6 * v) for k, v in sorted(per_centre.items())))
                if sc is not self.score_f:
                    self.score_f[self.own_z(z0, z1), y0:y1, x0:x1] = sc[z0 - o[0]:z1 - o[0], y0 - o[1]:y1 - o[1], x0 - o[2]:x1 - o[2]]
            if self.provider:
                with backend.host_timer("patch_bits"):
                    zz, yy, xx = torch.meshgrid(torch.arange(z0, z1, device=self.dev), torch.arange(y0, y1, device=self.dev),
                                                torch.arange(x0, x1, device=self.dev), indexing="ij")
                    cen = torch.stack([zz.reshape(-1), yy.reshape(-1), xx.reshape(-1)], 1).to(torch.int32)
                    row = ((zz - self.oz0) * self.Y + yy) * self.X + xx
                    self.bits_own[row.reshape(-1)] = self.ops.patch_bits(fr.pred, self.to_local(cen, fr), self.kw["fc_threshold"], self.params(fr))
                    del zz, yy, xx, cen, row
            if self.keep_cons:
                self.kept[t] = (cons, P, fr)
            del cons, sc, fr
        if not self.sharded and self.comm.world > 1:
            if self.rank_ranges is not None:
                self.comm.all_gather_slabs(self.score_f, self.rank_ranges)
            else:
                self.comm.all_reduce_sum(self.score_f)
        # The pooled consensus buffer stays allocated for the whole call (handing tens of GB back to
        # the driver and asking for them again costs seconds: 2 s + 1.3 s at 512^3); the global stage
        # carves its big temporaries -- the dense patch-bit table and the gathered candidate bits --
        # out of it.
        self.scratch = self.pool.view(torch.int32) if self.pool is not None else None


    def select_patches(self):
        """stage B: ranking and greedy cover (replicated, or sharded over the ranks by z)"""
        import torch
        def owned(z):
            m = torch.zeros_like(z, dtype=torch.bool)
            for (z0, z1) in self.my_slabs:
                m |= (z >= z0) & (z < z1)
            return m
        self.owned = owned

        def bits_of_own(coords_own, thresh, use_scratch=False):
            """patch bits of centres in the own slabs (global int32 [n, 3])"""
            if self.provider:
                c = coords_own.to(torch.int64)
                return self.bits_own[((c[:, 0] - self.oz0) * self.Y + c[:, 1]) * self.X + c[:, 2]]
            kw_s = {"scratch": self.scratch} if use_scratch and self.scratch is not None and hasattr(self.ops, "voxel_major_pool") else {}
            return self.ops.patch_bits(self.whole.pred, self.to_local(coords_own, self.whole), thresh, self.params(self.whole), **kw_s)
        self.bits_of_own = bits_of_own

        def gathered_bits(coords_t, thresh, use_scratch=False):
            """Patch bits of `coords_t` (global, int32 [n, 3] on the device): each rank packs those
            centred in its own slabs, the sum over ranks is the full table."""
            if self.comm.world == 1:
                return self.bits_of_own(coords_t, thresh, use_scratch)       # one rank owns everything
            bits = torch.zeros((int(coords_t.shape[0]), self.words), dtype=torch.int32, device=self.dev)
            mine = torch.nonzero(self.owned(coords_t[:, 0])).reshape(-1)
            if mine.numel():
                bits[mine] = self.bits_of_own(coords_t[mine], thresh)
            self.comm.all_reduce_sum(bits)
            return bits
        self.gathered_bits = gathered_bits

        def coords_of(lin_g):
            return torch.stack([lin_g // self.plane, (lin_g // self.X) % self.Y, lin_g % self.X], dim=1).to(torch.int32)

        def local_params(a, b):
            return backend.make_params((b - a, self.Y, self.X), self.ps, origin=(a, 0, 0), **self.flags)

        from .vote_instances import foreground_cover as fc
        pix_ths = fc._pix_thresholds(self.ps, self.kw)
        thr = self.kw.get("score_threshold", False)
        self.debug_crc = os.environ.get("PPP_DEBUG_CRC") == "1"     # development aid (tools/cover_repro.py)
        self.injected = self.kw.get("selected_patches") is not None
        if self.injected:
            self.sel_coords = np.array(list(self.kw["selected_patches"]), dtype=np.int32).reshape(-1, 3)
        elif self.sharded:
            # ---- stage B2: ranking and greedy cover SHARDED by z -- no rank holds a list as long as
            # the volume.  (i) every rank sorts the patches of its own range; (ii) the position of a
            # patch in the global ranked list = its own position + the number of patches of every
            # other rank that precede it -- a binary search in that rank's sorted scores (all-gathered:
            # 4 bytes per foreground voxel); ties keep raster order, i.e. lower ranks first;
            # (iii) the cover rounds run on the own slices with the global positions as priorities
            # (sharded_cover_own); (iv) the selected patches -- a short list -- are gathered.
            with backend.host_timer("sort"):
                fg_own = torch.zeros_like(self.fg_d)
                fg_own[self.own_z(self.oz0, self.oz1)] = self.fg_d[self.own_z(self.oz0, self.oz1)]
                lin_l, sc_own = self.ops.rank_order(self.score_f, fg_own, self.ps)       # field-frame indices, sorted
                del fg_own
                lin_own = lin_l + self.flo * self.plane
                del lin_l
                n_own = int(lin_own.numel())
                rank_id = torch.arange(n_own, dtype=torch.int64, device=self.dev)
                neg = -sc_own
                for r, other in enumerate(gather_lists(self.comm, neg)):
                    if r != self.comm.rank and other.numel():
                        rank_id += torch.searchsorted(other.contiguous(), neg, right=(r < self.comm.rank))
                del neg
                backend.note("ranked_own", n_own)
            del self.score_f
            with backend.host_timer("s3_cover"):
                if self.kw.get("skipSelection", False):
                    sel_own = torch.ones(n_own, dtype=torch.bool, device=self.dev)
                else:
                    never = torch.zeros(n_own, dtype=torch.bool, device=self.dev)
                    if self.any_overlap:
                        never |= self.ov_d.reshape(-1)[lin_own - self.flo * self.plane] != 0
                    if isinstance(thr, float):
                        never |= sc_own.double() < thr
                    h = self.ps[0] - 1
                    a, b = max(0, self.oz0 - h), min(self.Z, self.oz1 + h)
                    sel_own = sharded_cover_own(self.ops, self.comm, self.shape, self.ps, (self.oz0, self.oz1), self.rank_ranges,
                                                self.mask_d[self.own_z(a, b)].clone(), a, self.interior_count(self.mask_d),
                                                lin_own, rank_id.to(torch.int32), never,
                                                self.bits_of_own(coords_of(lin_own), self.kw["fc_threshold"], True),
                                                pix_ths, local_params)
                    del never
                mine_sel = torch.stack([rank_id[sel_own], lin_own[sel_own]], 1)
                allsel = torch.cat(gather_lists(self.comm, mine_sel), 0)
                allsel = allsel[torch.argsort(allsel[:, 0])]                  # global rank order
                self.sel_coords = coords_of(allsel[:, 1]).cpu().numpy()
                del mine_sel, allsel, sel_own, lin_own, sc_own, rank_id
        else:
            # ---- stage B: ranking, greedy cover (global; identical on every rank) ---------------
            # The ranked list stays on the device (it has one entry per foreground voxel of the
            # GLOBAL volume); only the selected patches come back to the host.
            if self.debug_crc:
                backend.note("crc_scores", zlib.crc32(self.score_f.cpu().numpy().tobytes()))
            with backend.host_timer("sort"):
                lin_t, rscores_t = self.ops.rank_order(self.score_f, self.fg_d, self.ps)
            if self.debug_crc:
                backend.note("crc_ranked", zlib.crc32(lin_t.cpu().numpy().tobytes()))
            scores_host = self.score_f.cpu().numpy() if self.seq_cover and self.kw.get("select_patches_overlap_neighborhood") else None
            del self.score_f
            self.coords_t = coords_of(lin_t)
            if self.kw.get("skipSelection", False):
                self.sel_coords = self.coords_t.cpu().numpy()
            elif self.seq_cover:
                # ---- sequential native cover with marks (one rank; the ranked list goes to the host,
                # the patch bits of a chunk of centres come from wherever the prediction lives)
                from .vote_instances.ranked_patches import PatchList
                with backend.host_timer("s3_cover"):
                    radslice = tuple(slice(int(self.rad[i]), self.shape[i] - int(self.rad[i])) for i in range(3))

                    def bits_of(coords):
                        c = torch.from_numpy(np.ascontiguousarray(coords, dtype=np.int32)).to(self.dev)
                        return self.gathered_bits(c, self.kw["fc_threshold"]).cpu().numpy().view(np.uint32)
                    ranked_h = PatchList(self.coords_t.cpu().numpy(), rscores_t.cpu().numpy())
                    sel_list, _ = fc.cover_sequential(self.ov_d.cpu().numpy(), self.mask_d.cpu().numpy() != 0, self.ps, ranked_h, radslice,
                                                      bits_of, scores_host, **self.kw)
                    self.sel_coords = np.ascontiguousarray(sel_list.coords, dtype=np.int32).reshape(-1, 3)
                del ranked_h, scores_host
            else:
                # sharded over the ranks when every rank owns one contiguous z-range
                # (PPP_COVER_SHARDED=0: every rank runs the whole cover; "force": also with one rank)
                shard_env = os.environ.get("PPP_COVER_SHARDED", "1")
                use_shard = (self.comm.world > 1 or shard_env == "force") and hasattr(self.ops, "cover_shard") and \
                    self.kw.get("_shard_cover", shard_env != "0") and self.rank_ranges is not None and \
                    all(r[1] - r[0] >= 2 * (self.ps[0] - 1) for r in self.rank_ranges)
                with backend.host_timer("s3_cover"):
                    # patches the loop never looks at (foreground_cover.py:136-141): centre on an
                    # overlap voxel; everything from the first score below score_threshold on
                    never = torch.zeros(lin_t.shape, dtype=torch.bool, device=self.dev)
                    if self.any_overlap:
                        never |= self.ov_d.reshape(-1)[lin_t] != 0
                    if isinstance(thr, float):
                        below = torch.nonzero(rscores_t.double() < thr).reshape(-1)
                        if below.numel():
                            never[int(below[0].item()):] = True
                    radslice = tuple(slice(int(self.rad[i]), self.shape[i] - int(self.rad[i])) for i in range(3))
                    if use_shard:
                        selected = sharded_cover(self.ops, self.comm, self.shape, self.ps, self.rank_ranges[self.comm.rank], self.rank_ranges,
                                                 self.mask_d, lin_t, never, pix_ths, radslice,
                                                 lambda idx: self.bits_of_own(self.coords_t[idx], self.kw["fc_threshold"]),
                                                 local_params)
                    elif self.provider and self.comm.world == 1 and hasattr(self.ops, "voxel_major_pool"):
                        # the per-voxel bit table of stage A serves the cover as it is (a copy in rank
                        # order would double its 92 bytes per voxel)
                        selected = self.ops.greedy_cover(self.mask_d, self.bits_own, lin_t, rscores_t, never, pix_ths, radslice,
                                                    self.Pg, self.kw, bits_first_voxel=self.oz0 * self.plane)
                    else:
                        bits = self.gathered_bits(self.coords_t, self.kw["fc_threshold"], True)
                        selected = self.ops.greedy_cover(self.mask_d, bits, lin_t, rscores_t, never, pix_ths, radslice, self.Pg, self.kw)
                        del bits
                    del never
                    self.sel_coords = self.coords_t[selected].cpu().numpy()
            del lin_t, rscores_t, self.coords_t

    def thin_cover(self):
        """set-cover thinning of the selected patches (replicated)"""
        import torch
        if self.debug_crc and not self.injected:
            backend.note("crc_selected", zlib.crc32(np.ascontiguousarray(self.sel_coords).tobytes()))
            if os.environ.get("PPP_STOP_AFTER_COVER") == "1":
                return self.early()
        backend.note("n_cover", len(self.sel_coords))
        if not self.kw.get("skipThinCover") and len(self.sel_coords) > 0:
            if self.kw.get("sample", 1.0) < 1.0:
                raise NotImplementedError("sample < 1 uses unseeded random sampling in the reference")
            h_t = self.ps[0] - 1
            shard_thin = self.comm.world > 1 and self.rank_ranges is not None and self.contiguous and \
                hasattr(self.ops, "thin_shard") and self.ps[2] <= 32 and \
                self.kw.get("_shard_thin", os.environ.get("PPP_THIN_SHARDED", "1") != "0") and \
                all(r[1] - r[0] >= 2 * h_t for r in self.rank_ranges)
            if shard_thin:
                # ---- sharded by z (round 6): every rank thins the selected patches of its own slices; the
                # slab-boundary zones travel point to point every round, nothing of the volume's size is
                # gathered (sharded_thin_own)
                with backend.host_timer("s4_thin"):
                    idx_own = np.flatnonzero((self.sel_coords[:, 0] >= self.oz0) & (self.sel_coords[:, 0] < self.oz1))
                    c_own = torch.from_numpy(np.ascontiguousarray(self.sel_coords[idx_own])).to(self.dev)
                    lin_own = (c_own[:, 0].to(torch.int64) * self.Y + c_own[:, 1]) * self.X + c_own[:, 2]
                    bits = self.bits_of_own(c_own, self.kw["fc_threshold"]) if len(idx_own) else \
                        torch.zeros((0, self.words), dtype=torch.int32, device=self.dev)
                    a_t, b_t = max(0, self.oz0 - h_t), min(self.Z, self.oz1 + h_t)
                    keep_idx, _ = sharded_thin_own(
                        self.ops, self.comm, self.shape, self.ps, (self.oz0, self.oz1), self.rank_ranges,
                        self.mask_d[self.own_z(a_t, b_t)].clone(), a_t, self.interior_count(self.mask_d),
                        lin_own, torch.from_numpy(idx_own.astype(np.int64)).to(self.dev), bits,
                        lambda a, b: backend.make_params((b - a, self.Y, self.X), self.ps, origin=(a, 0, 0), **self.flags))
                    self.sel_coords = self.sel_coords[keep_idx.cpu().numpy()]
                    del bits, c_own, lin_own, keep_idx
            else:
                with backend.host_timer("s4_thin"):
                    # replicated: every rank thins the same (short) global list on its own device; with
                    # local fields the mask of the whole volume is gathered for it (one byte per voxel)
                    if self.any_local:
                        mask_g = torch.zeros(self.shape, dtype=torch.uint8, device=self.dev)
                        mask_g[self.oz0:self.oz1] = self.mask_d[self.own_z(self.oz0, self.oz1)]
                        self.comm.all_gather_slabs(mask_g, self.rank_ranges)
                    else:
                        mask_g = self.mask_d
                    sel_t = torch.from_numpy(np.ascontiguousarray(self.sel_coords)).to(self.dev)
                    bits = self.gathered_bits(sel_t, self.kw["fc_threshold"])
                    sel_lin = (self.sel_coords[:, 0].astype(np.int64) * self.Y + self.sel_coords[:, 1]) * self.X + self.sel_coords[:, 2]
                    if hasattr(self.ops, "thin_cover") and os.environ.get("PPP_THIN", "device") != "host" \
                            and self.ps[2] <= 32:
                        keep = self.ops.thin_cover(mask_g, bits, torch.from_numpy(sel_lin).to(self.dev), self.Pg)
                        keep = keep.cpu().numpy()
                    else:
                        keep = backend.host_thin_cover(np.ascontiguousarray(mask_g.cpu().numpy()),
                                                       self.ps, np.ascontiguousarray(sel_lin),
                                                       bits.cpu().numpy().view(np.uint32))
                    self.sel_coords = self.sel_coords[keep]
                    del bits, sel_t, mask_g
        self.bits_own = None


    def pair_affinities(self):
        """pair enumeration and stage C: the pair affinities, per tile"""
        import torch
        # ---- pairs (global coordinates; replicated, it is cheap) -------------------------------
        order = np.argsort(self.sel_coords[:, 2], kind="stable")
        self.nodes = np.ascontiguousarray(self.sel_coords[order].astype(np.int32))
        self.nodes_dev = torch.from_numpy(self.nodes).to(self.dev)
        max_ps = self.kw.get("max_total_patch_distance_in_ps_multiples", 2)
        streaming = not self.want_inter and not self.kw.get("mws") and self.kw.get("selected_patch_pairs") is None \
            and self.kw.get("_stream_pairs", os.environ.get("PPP_STREAM_PAIRS", "1") != "0") \
            and hasattr(self.ops, "label_state")

        pairs_frame_box = self.pairs_frame_box

        def next_of(t):
            i = self.my_tiles.index(t)
            return self.my_tiles[i + 1] if i + 1 < len(self.my_tiles) else None
        self.next_of = next_of

        def tile_consensus(t):
            """(frame, cons, P) for the pairs whose patch A lies in tile t"""
            if self.keep_cons:
                cons, P, fr = self.kept.pop(t)
                return fr, cons, P
            cbox = self.bases_for_pairs(t)
            if self.cache is not None:
                cons, P = self.rows_from_cache(self.whole, cbox, self.pool)
                return self.whole, cons, P
            if self.ring_z:
                cons, P = self.ring_rows(self.whole, t, True, self.pool)
                return self.whole, cons, P
            nxt = self.next_of(t)
            fr = self.frame_for(self.pairs_frame_box(t), self.pairs_frame_box(nxt) if nxt is not None else None)
            cons, P = self.consensus_of(fr, self.params(fr, cbox), self.pool)
            return fr, cons, P

        self.state = None
        if streaming:
            # ---- stage C (streaming): the pair rows of a tile are enumerated, scored and fed to
            # the union-find while the tile's consensus is alive; they are never gathered.  A row
            # belongs to the tile (hence the rank) of its first patch; its GLOBAL row id -- the
            # position it would have in the canonical list -- comes from the exclusive scan of the
            # per-patch partner counts, which every rank computes for its own patches only.
            def tile_subset(t):
                z0, z1, y0, y1, x0, x1 = t
                own = (self.nodes_dev[:, 0] >= z0) & (self.nodes_dev[:, 0] < z1)
                if self.ny_t > 1 or self.nx_t > 1:
                    own &= (self.nodes_dev[:, 1] >= y0) & (self.nodes_dev[:, 1] < y1) & \
                           (self.nodes_dev[:, 2] >= x0) & (self.nodes_dev[:, 2] < x1)
                return torch.nonzero(own).reshape(-1)

            subsets = [tile_subset(t) for t in self.my_tiles]
            with backend.host_timer("pairs"):
                counts = self.ops.pair_counts(self.nodes_dev, torch.cat(subsets) if subsets else
                                         torch.zeros((0,), dtype=torch.int64, device=self.dev), self.Pg, max_ps)
                self.comm.all_reduce_sum(counts)
                ends = torch.cumsum(counts, 0)
                n_pair_rows = int(ends[-1].item()) if len(self.nodes) else 0
                goffsets = (ends - counts).contiguous()
                del ends
            n_rows = n_pair_rows + (len(self.nodes) if self.kw["includeSinglePatchCCS"] else 0)
            if n_rows == 0:
                return self.early()
            backend.note("n_selected", len(self.nodes))
            backend.note("n_pairs", n_rows)
            self.state = self.ops.label_state(self.nodes_dev, self.Pg)
            with backend.host_timer("s5_patch_graph"):
                for t, subset in zip(self.my_tiles, subsets):
                    with backend.host_timer("s5a_select_rows"):
                        rows_t, gid_t = self.ops.pairs_subset(self.nodes_dev, subset, counts, goffsets, n_pair_rows,
                                                         self.Pg, max_ps, self.kw["includeSinglePatchCCS"])
                        if rows_t is None:
                            continue
                    with backend.host_timer("s5b_consensus"):
                        fr, cons, P = tile_consensus(t)
                    with backend.host_timer("s5c_patch_graph"):
                        a = self.ops.patch_graph(fr.pred, cons, self.to_local(rows_t, fr, 2), P)
                    with backend.host_timer("s6_label_paint"):
                        self.state.add(rows_t, a, gid_t)
                    del cons, rows_t, gid_t, a, fr
            self.kept.clear()
            del counts, goffsets, subsets
            if self.comm.world > 1:
                # boundary-label merge: every rank's forest (node -> parent) is gathered and united
                # with the own one; first appearances / "has a positive edge" are reduced
                par, fp, hp = self.state.export()
                self.state.merge(self.comm.all_gather(par), self.comm.all_reduce_min(fp), self.comm.all_reduce_max(hp))
                del par, fp, hp
            self.rows = None
        elif self.kw.get("selected_patch_pairs") is not None:
            rows_host = np.ascontiguousarray(
                np.array(self.kw["selected_patch_pairs"], dtype=np.uint32).reshape(-1, 6))
            self.rows = torch.from_numpy(rows_host.view(np.int32)).to(self.dev) if len(rows_host) else None
        else:
            with backend.host_timer("pairs"):
                self.rows = self.ops.patch_pairs(self.nodes_dev, self.Pg, max_ps, self.kw["includeSinglePatchCCS"])
        if self.state is None and self.rows is None:
            return self.early()
        if self.state is None:
            n_rows = int(self.rows.shape[0])
            backend.note("n_selected", len(self.nodes))
            backend.note("n_pairs", n_rows)

        # ---- stage C (materialised list: intermediates wanted, injected pairs, mutex watershed):
        # pair affinities, each pair on the rank / tile that owns patch A
        self.aff = None if self.state is not None else torch.zeros((n_rows,), dtype=torch.float32, device=self.dev)
        with backend.host_timer("s5_patch_graph"):
            rows_of_tile = None
            if self.state is None and len(self.my_tiles) > 1:
                with backend.host_timer("s5a_select_rows"):
                    rows_of_tile = rows_by_tile(self.rows, self.my_tiles)
            for n_t, t in enumerate(self.my_tiles if self.state is None else []):
                z0, z1, y0, y1, x0, x1 = t
                with backend.host_timer("s5a_select_rows"):
                    if rows_of_tile is not None:
                        idx = rows_of_tile.pop(n_t)
                    else:
                        own = (self.rows[:, 0] >= z0) & (self.rows[:, 0] < z1)
                        if self.ny_t > 1 or self.nx_t > 1:
                            own &= (self.rows[:, 1] >= y0) & (self.rows[:, 1] < y1) & \
                                   (self.rows[:, 2] >= x0) & (self.rows[:, 2] < x1)
                        idx = torch.nonzero(own).reshape(-1)
                        del own
                    if idx.numel() == 0:
                        continue
                with backend.host_timer("s5b_consensus"):
                    fr, cons, P = tile_consensus(t)
                with backend.host_timer("s5c_patch_graph"):
                    a = self.ops.patch_graph(fr.pred, cons, self.to_local(self.rows[idx], fr, 2), P)
                with backend.host_timer("s5d_scatter"):
                    self.aff[idx] = a
                del cons, idx, a, fr
        self.kept.clear()
        self.pool = self.scratch = self.cache = None
        if self.state is None:
            self.comm.all_reduce_sum(self.aff)
        if self.want_inter:
            return self.rows.cpu().numpy().view(np.uint32), self.aff.cpu().numpy()


    def label_and_paint(self):
        """stage D: components / mutex watershed and the painting of the own slabs"""
        import torch
        # ---- stage D: components (replicated) and painting of the own slabs ---------------------
        with backend.host_timer("s6_label_paint"):
            if self.kw.get("mws") and self.kw.get("selected_patch_pairs") is None and hasattr(self.ops, "mws_labels"):
                # the library's own pair list never repeats a node pair: edge order and |aff| sort on
                # the device, only the sequential loop on the host -- on rank 0 alone (the loop is one
                # host thread; N copies of it on one node's memory system run slower than one), the
                # labels reach the others as one SUM all-reduce of 4 bytes per selected patch
                if self.comm.world > 1 and os.environ.get("PPP_MWS_RANK0", "1") != "0":
                    if self.comm.rank == 0:
                        lab_all, n_labels = self.ops.mws_labels(self.rows, self.aff, self.nodes_dev, self.Pg)
                        lab_all = torch.cat([lab_all.to(torch.int32),
                                             torch.tensor([n_labels], dtype=torch.int32, device=self.dev)])
                    else:
                        lab_all = torch.zeros((len(self.nodes) + 1,), dtype=torch.int32, device=self.dev)
                    self.comm.all_reduce_sum(lab_all)
                    n_labels = int(lab_all[-1].item())
                    lab_all = lab_all[:-1]
                else:
                    lab_all, n_labels = self.ops.mws_labels(self.rows, self.aff, self.nodes_dev, self.Pg)
                held = lab_all > 0
                lab_nodes, labels = self.nodes_dev[held], lab_all[held]
                del lab_all, held
            elif self.kw.get("mws"):
                lab_nodes, labels, n_labels = backend.host_mws(self.rows.cpu().numpy().view(np.uint32),
                                                               self.aff.cpu().numpy(), self.shape)
                lab_nodes = torch.from_numpy(np.ascontiguousarray(lab_nodes)).to(self.dev)
                labels = torch.from_numpy(labels.astype(np.int32)).to(self.dev)
            else:
                if self.state is not None:
                    keys = self.state.finish()
                    valid = keys != backend.NONE_KEY64
                    self.state = None
                else:
                    keys = self.ops.label_components(self.rows, self.aff, self.nodes_dev, self.Pg)
                    valid = keys != backend.NONE_KEY
                # component ids in the order of their keys (= networkx's component order)
                uniq, inverse = torch.unique(keys[valid], sorted=True, return_inverse=True)
                lab_nodes = self.nodes_dev[valid]
                labels = (inverse + 1).to(torch.int32)
                n_labels = int(uniq.numel())
                del keys, valid, uniq, inverse
            del self.rows, self.aff
            # ids are uint16 in the whole-volume entry (vote_instances.py:230; its np.seterr(over=
            # 'raise') makes an id above 65 535 an error) and uint32 in the blockwise / stitched one
            # (stitch_patch_graph.py:120), which the caller asks for with _instances_dtype
            if n_labels > np.iinfo(self.id_dtype).max:
                raise OverflowError("%d instance ids do not fit %s (the blockwise entry, "
                                    "stitch_patch_graph.main, carries uint32 ids)" % (n_labels, self.id_dtype.name))
            backend.note("ids_issued", n_labels)
            # painted on the own slabs only; `inst_g` is the whole map when it is gathered
            gz0, gz1 = (0, self.Z) if self.gather_result else (self.oz0, self.oz1)
            inst_g = torch.zeros((gz1 - gz0, self.Y, self.X), dtype=torch.int32, device=self.dev)
            if self.whole is not None:
                Pl = self.params(self.whole)
                for (z0, z1) in self.my_slabs:
                    near = torch.nonzero((lab_nodes[:, 0] >= z0 - self.rz) & (lab_nodes[:, 0] < z1 + self.rz)).reshape(-1)
                    if near.numel() == 0:
                        continue
                    inst_l = torch.zeros(self.whole.shape, dtype=torch.int32, device=self.dev)
                    self.ops.paint(self.whole.pred, self.to_local(lab_nodes[near], self.whole), labels[near].contiguous(), inst_l, Pl)
                    inst_g[z0 - gz0:z1 - gz0] = inst_l[z0 - self.lo:z1 - self.lo]
                    del inst_l
            else:
                # nodes sorted by z once: a tile looks at the nodes of its own z-range (+ radius) only
                # instead of scanning the whole list (640 tiles x 3.6 M nodes at 1024^3); the painting
                # keeps the largest label per voxel, so the order of the nodes does not matter
                if len(self.my_tiles) > 1 and lab_nodes.shape[0] > 0:
                    z_order = torch.argsort(lab_nodes[:, 0].contiguous(), stable=True)
                    lab_nodes, labels = lab_nodes[z_order].contiguous(), labels[z_order].contiguous()
                    node_z = lab_nodes[:, 0].contiguous()
                    del z_order
                else:
                    node_z = None
                for t in self.my_tiles:
                    z0, z1, y0, y1, x0, x1 = t
                    if node_z is not None:
                        za = int(torch.searchsorted(node_z, torch.tensor([z0 - self.rz], dtype=node_z.dtype, device=self.dev)).item())
                        zb = int(torch.searchsorted(node_z, torch.tensor([z1 + self.rz], dtype=node_z.dtype, device=self.dev)).item())
                        part = lab_nodes[za:zb]
                        near = (part[:, 1] >= y0 - int(self.rad[1])) & (part[:, 1] < y1 + int(self.rad[1]))
                        near &= (part[:, 2] >= x0 - int(self.rad[2])) & (part[:, 2] < x1 + int(self.rad[2]))
                        near = torch.nonzero(near).reshape(-1) + za
                        del part
                    else:
                        near = (lab_nodes[:, 0] >= z0 - self.rz) & (lab_nodes[:, 0] < z1 + self.rz)
                        near &= (lab_nodes[:, 1] >= y0 - int(self.rad[1])) & (lab_nodes[:, 1] < y1 + int(self.rad[1]))
                        near &= (lab_nodes[:, 2] >= x0 - int(self.rad[2])) & (lab_nodes[:, 2] < x1 + int(self.rad[2]))
                        near = torch.nonzero(near).reshape(-1)
                    if near.numel() == 0:
                        continue
                    nxt = self.next_of(t)
                    fr = self.frame_for(self.grow(t, self.rad), self.grow(nxt, self.rad) if nxt is not None else None)
                    o = fr.origin
                    inst_l = torch.zeros(fr.shape, dtype=torch.int32, device=self.dev)
                    self.ops.paint(fr.pred, self.to_local(lab_nodes[near], fr), labels[near].contiguous(), inst_l, self.params(fr))
                    inst_g[z0 - gz0:z1 - gz0, y0:y1, x0:x1] = \
                        inst_l[z0 - o[0]:z1 - o[0], y0 - o[1]:y1 - o[1], x0 - o[2]:x1 - o[2]]
                    del inst_l, fr
            if not self.gather_result:
                instances = inst_g.cpu().numpy().view(np.uint32).astype(self.id_dtype, copy=False)
                return instances, self.fg_out(self.oz0, self.oz1)
            if self.id_dtype == np.uint32:
                if self.comm.world > 1:
                    if self.rank_ranges is not None:
                        self.comm.all_gather_slabs(inst_g, self.rank_ranges)
                    else:
                        self.comm.all_reduce_sum(inst_g)
                instances = inst_g.cpu().numpy().view(np.uint32)
            else:
                # ids fit 16 bits (checked above): half the bytes on the wire and to the host
                inst16 = inst_g.to(torch.int16)
                del inst_g
                if self.comm.world > 1:
                    if self.rank_ranges is not None:
                        self.comm.all_gather_slabs(inst16, self.rank_ranges)
                    else:
                        inst32 = inst16.to(torch.int32) & 0xFFFF
                        self.comm.all_reduce_sum(inst32)
                        inst16 = inst32.to(torch.int16)
                instances = inst16.cpu().numpy().view(np.uint16)
        return instances, self.full_fg()


def slabs_needed(shape, patchshape, free_bytes, safety=0.6, copies=3.0):
    """Smallest number of z-slabs whose consensus working set (`copies` x the compact size: 3 =
    compact planes + the voxel-major copy ranking and patch graph read, 2 = voxel-major written
    directly by S1; on a box of own thickness + 4*rz slices) fits into `free_bytes` of device
    memory."""
    pz, py, px = [int(p) for p in patchshape]
    planes = ((2 * pz - 1) * (2 * py - 1) * (2 * px - 1) - 1) // 2
    per_slice = float(copies) * planes * 4 * int(shape[1]) * int(shape[2])
    extra = 4 * (pz // 2)
    budget = safety * free_bytes
    Z = int(shape[0])
    for n in range(1, Z + 1):
        own = -(-Z // n)
        if (min(Z, own + extra)) * per_slice <= budget:
            return n
    return Z


def tiles_needed(shape, patchshape, free_bytes, safety=0.6, copies=3.0):
    """(n_slabs, ny, nx): the grid of tiles with the least consensus work among those whose
    working set -- `copies` x the compact consensus on the tile grown by the pairs halo (3 =
    compact planes + the voxel-major copy, 2 = voxel-major written directly by S1) -- fits
    `safety * free_bytes`.  Work = the base voxels of both passes (scores: tile + radius; pairs:
    tile + radius + p - 1 on the low side in z and on both sides in y / x), clipped to the volume:
    cube-like tiles win, thin slabs spend most of their work on the halo.  (1, 1, 1) when the
    whole volume fits."""
    pz, py, px = [int(p) for p in patchshape]
    planes = ((2 * pz - 1) * (2 * py - 1) * (2 * px - 1) - 1) // 2
    Z, Y, X = [int(v) for v in shape]
    budget = safety * free_bytes
    per_voxel = float(copies) * planes * 4

    def axis(n_vox, n_tiles, p, both):
        """(largest pairs-box extent, sum of pairs-box extents, sum of scores-box extents) of
        n_tiles tiles along an axis of n_vox voxels"""
        r, g = p // 2, p - 1
        big = s_pairs = s_scores = 0
        for (a, b) in plan_slabs(n_vox, n_tiles):
            e = min(n_vox, b + r + (g if both else 0)) - max(0, a - r - g)
            big = max(big, e)
            s_pairs += e
            s_scores += min(n_vox, b + r) - max(0, a - r)
        return big, s_pairs, s_scores

    if per_voxel * Z * Y * X <= budget:
        return 1, 1, 1
    best = None
    zs = [axis(Z, n, pz, False) for n in range(1, Z + 1)]
    ys = [axis(Y, n, py, True) for n in range(1, min(Y, 64) + 1)]
    xs = [axis(X, n, px, True) for n in range(1, min(X, 64) + 1)]
    for n, (bz, pz_sum, sz_sum) in enumerate(zs, 1):
        for ny, (by, py_sum, sy_sum) in enumerate(ys, 1):
            # smallest nx that fits (more cuts only add halo)
            nx = next((k for k, (bx, _, _) in enumerate(xs, 1) if per_voxel * bz * by * bx <= budget), None)
            if nx is None:
                continue
            _, px_sum, sx_sum = xs[nx - 1]
            work = float(pz_sum) * py_sum * px_sum + float(sz_sum) * sy_sum * sx_sum
            if best is None or work < best[0] - 1e-9:
                best = (work, n, ny, nx)
    if best is None:
        return Z, min(Y, 64), min(X, 64)
    return best[1], best[2], best[3]


def pairs_box_voxels(shape, patchshape, n, ny, nx):
    """base voxels of the largest pairs box (tile + radius + p - 1 below in z, on both sides in
    y / x, clipped to the volume) of an n x ny x nx grid of tiles"""
    big = 1
    for ext, k, p, both in zip(shape, (n, ny, nx), patchshape, (False, True, True)):
        r, g = int(p) // 2, int(p) - 1
        big *= max(min(int(ext), b + r + (g if both else 0)) - max(0, a - r - g) for a, b in plan_slabs(int(ext), k))
    return big


def cons_cache_bytes(box_shape, patchshape):
    """bytes of the COMPACT consensus planes over a box of base voxels"""
    pz, py, px = [int(p) for p in patchshape]
    planes = ((2 * pz - 1) * (2 * py - 1) * (2 * px - 1) - 1) // 2
    return 4.0 * planes * float(np.prod([int(v) for v in box_shape]))


def plan_tiles(own_shape, patchshape, free_bytes, safety=0.6, copies=3.0, cache_shape=None, min_pool=10e9):
    """(n_slabs, ny, nx, use_cache).  With `cache_shape` -- the box of base voxels a consensus cache
    would span: the rank's own block grown by the pairs halo, clipped -- the cache is taken when its
    planes fit and at least `min_pool` bytes remain for the rows of one tile; the tile grid is then
    planned for what is left.  PPP_CONS_CACHE=0 / 1 overrides the memory rule (1: taken whenever a
    grid still fits)."""
    mode = os.environ.get("PPP_CONS_CACHE", "auto")
    budget = safety * free_bytes
    whole_fits = tiles_needed(own_shape, patchshape, free_bytes, safety=safety, copies=copies) == (1, 1, 1)
    if cache_shape is not None and mode != "0" and not whole_fits:
        left = budget - cons_cache_bytes(cache_shape, patchshape)
        if left >= (min_pool if mode != "1" else 1e9):
            n, ny, nx = tiles_needed(own_shape, patchshape, left, safety=1.0, copies=copies)
            if float(copies) * cons_cache_bytes((1, 1, 1), patchshape) * pairs_box_voxels(own_shape, patchshape, n, ny, nx) <= left:
                return n, ny, nx, True       # (tiles_needed's last resort may not fit)
    return tiles_needed(own_shape, patchshape, free_bytes, safety=safety, copies=copies) + (False,)


def consensus_work(shape, patchshape, n, ny, nx, ring=False):
    """base voxels S1 computes for an n x ny x nx grid of tiles, both passes (scores: tile + radius;
    pairs: tile + radius + p - 1 below in z and on both sides in y / x; clipped to the volume).
    ring: the z-sweep with the rows in a ring -- no z-halo, only the p - 1 source slices below the
    first tile of a column in the pairs pass."""
    ext = []
    for e, k, p, both in zip(shape, (n, ny, nx), patchshape, (False, True, True)):
        r, g = int(p) // 2, int(p) - 1
        sc = pa = 0
        for a, b in plan_slabs(int(e), k):
            sc += min(int(e), b + r) - max(0, a - r)
            pa += min(int(e), b + r + (g if both else 0)) - max(0, a - r - g)
        ext.append((sc, pa))
    if ring:
        z_sc = z_pa = int(shape[0])        # (the first tile starts at slice 0 of the block: nothing below it)
    else:
        z_sc, z_pa = ext[0]
    return float(z_sc) * ext[1][0] * ext[2][0] + float(z_pa) * ext[1][1] * ext[2][1]


def ring_margin(pz):
    """Slices a ring of rows needs beyond a tile's thickness: the radius on both sides, the pairs
    pass's p - 1 source slices below and p - 1 mirrored slices above, four spare."""
    pz = int(pz)
    return 2 * (pz // 2) + 2 * (pz - 1) + 4


def plan_ring(own_shape, patchshape, free_bytes, safety=0.6, copies=2.0, min_thick=16, gain=0.95):
    """(n_slabs, ny, nx, ring_z) for the z-sweep with a ring of rows (tiling.assemble, `_ring_z`), or
    None when it does not pay: columns of ny x nx tiles in y / x, the ring as many slices as the
    budget holds of a column's pairs box, tiles a multiple of 8 slices thick (the ranking kernel's
    tiles of centres are 8 thick) with ring >= thick + ring_margin(pz) (28 at pz = 9: what
    assemble() asks of `_ring_z`); taken when S1's work falls below `gain` x that of the best plain
    grid.  PPP_RING=0 switches it off."""
    if os.environ.get("PPP_RING", "1") == "0":
        return None
    pz, py, px = [int(p) for p in patchshape]
    Z, Y, X = [int(v) for v in own_shape]
    row_bytes = float(copies) * cons_cache_bytes((1, 1, 1), patchshape)
    budget = safety * free_bytes
    plain = tiles_needed(own_shape, patchshape, free_bytes, safety=safety, copies=copies)
    if plain == (1, 1, 1) or pz < 3:
        return None
    def rank_tail(n, ny, nx):
        """how well a ranking launch of one tile fills the chip: workgroups of 8 x 8 x 16 centres,
        four resident per CU -- a launch of 1.5 x 1024 workgroups takes as long as one of 2 x 1024"""
        wgs = max(-(-(b - a) // 8) for a, b in plan_slabs(Z, n)) * -(-(-(-Y // ny)) // 8) * -(-(-(-X // nx)) // 16)
        return wgs / (-(-wgs // 1024) * 1024.0)

    def cost(n, ny, nx, ring):
        # S1 work per voxel of both passes + the ranking's share (11.0 of 25.9 s at 512^3 / 9^3,
        # profiles/r04_zv: 1.1 x the cost of one S1 pass per voxel) over how well its launches fill
        return consensus_work(own_shape, patchshape, n, ny, nx, ring=ring) / float(Z * Y * X) + 1.1 / rank_tail(n, ny, nx)

    best = None
    margin = ring_margin(pz)
    for ny in range(1, min(Y, 8) + 1):
        for nx in range(1, min(X, 8) + 1):
            area = pairs_box_voxels((1, Y, X), (1, py, px), 1, ny, nx)
            slices = int(budget // (row_bytes * area))
            top = min((slices - margin) // 8 * 8, -(-Z // 8) * 8)
            for thick in range(top, max(min_thick, pz - 1, 8) - 1, -8):
                n = -(-Z // thick)
                if n < 2:
                    continue
                key = (cost(n, ny, nx, True), n * ny * nx)
                if best is None or key < best[0]:
                    best = (key, n, ny, nx, max(b - a for a, b in plan_slabs(Z, n)) + margin)
    if best is None or best[0][0] > gain * cost(*plain, False):
        return None
    return best[1], best[2], best[3], best[4]


def dry_run_plan(shape, patchshape, world, hbm_gb=309.2, usable=0.985, provider=False, halo_mode="exchange",
                 cover_frac=0.0346, thin_frac=0.00339, pairs_per_voxel=0.291, cover_rounds=700, thin_rounds=40,
                 result_gather=False):
    """What `world` ranks would hold and move for one volume -- WITHOUT a GPU (bench.py --dry-run-plan).
    Per rank: z-range, halo, resident prediction bytes, the tile / ring / cache plan the memory rule
    takes for the HBM left, S1 work per owned voxel; per step and rank: bytes received in every
    collective of tiling.assemble.  The list densities default to the measured dense synthetic
    512^3 / 9^3 step (profiles/r05_zzg_*: 4.64 M cover patches, 454 k after thinning, 39.0 M pair rows
    per 134 M voxels, ~700 cover rounds)."""
    Z, Y, X = [int(v) for v in shape]
    ps = [int(p) for p in patchshape]
    C, H, plane = int(np.prod(ps)), halo(ps), Y * X
    W = (2 * ps[0] - 1) * (2 * ps[1] - 1) * (2 * ps[2] - 1)
    words = (C + 31) // 32
    V = float(Z) * plane
    out = {"volume": [Z, Y, X], "patchshape": ps, "ranks": int(world), "halo_slices": H,
           "hbm_gb_assumed": hbm_gb, "usable_fraction": usable, "ranks_plan": [], "assumed_densities": {
               "cover_patches_per_voxel": cover_frac, "thinned_patches_per_voxel": thin_frac,
               "pair_rows_per_voxel": pairs_per_voxel, "cover_rounds": cover_rounds, "thin_rounds": thin_rounds}}
    slabs = plan_slabs(Z, world)
    for r in range(world):
        mine = slabs_of_rank(slabs, r, world)
        if not mine:
            continue
        oz0, oz1 = mine[0][0], mine[-1][1]
        lo, hi = local_range(mine, Z, ps)
        own_v, held_v = float(oz1 - oz0) * plane, float(hi - lo) * plane
        pred_own = 2.0 * C * own_v
        pred_held = 0.0 if provider else 2.0 * C * held_v
        resident = pred_held          # (halo exchange: in place, into the halo slices of the same buffer)
        tile_pred = 2.0 * C * 180 ** 3 if provider else 0.0
        # (the same reserves as the callers' memory plans: vote_instances.to_instance_seg on one rank,
        # bench.py's multi-rank workloads; 288 GiB of HBM3E = 309.2 GB, ~1.5 % held by the runtime)
        fields = (70.0 * held_v + 4e9) if world == 1 else (150.0 * held_v + 6e9)
        free = usable * hbm_gb * 1e9 - resident - tile_pred
        budget = max(free - fields, 0.25 * free)
        own_shape = (oz1 - oz0, Y, X)
        cz0, cz1 = max(0, oz0 - ps[0] // 2 - (ps[0] - 1)), min(Z, oz1 + ps[0] // 2)
        n, ny, nx, cache = plan_tiles(own_shape, ps, budget, safety=0.92, copies=2.0,
                                      cache_shape=None if provider else (cz1 - cz0, Y, X))
        ring = None
        if not cache and not provider:
            ring = plan_ring(own_shape, ps, budget, safety=0.92, copies=2.0)
            if ring is not None:
                n, ny, nx = ring[:3]
        work = consensus_work(own_shape, ps, n, ny, nx, ring=ring is not None) / own_v
        if cache:
            work = float(cz1 - cz0) / (oz1 - oz0)
        nb = (1 if oz0 > 0 else 0) + (1 if oz1 < Z else 0)
        zone = 2 * (ps[0] - 1) * plane                    # voxels of one boundary zone
        recv = {
            "prediction_halo": 2.0 * C * (held_v - own_v) if (halo_mode == "exchange" and not provider and world > 1) else 0.0,
            "field_halo": 3.0 * (held_v - own_v) if (halo_mode == "exchange" and world > 1) else 0.0,
            "sorted_scores_all_gather": 4.0 * (V - own_v) if world > 1 else 0.0,
            # per round: rank volume (int32) after the count step, mask + dirty marks (bytes) after select
            "cover_zones_point_to_point": float(cover_rounds) * nb * zone * (4.0 + 1.0 + 1.0),
            "cover_selected_gather": 16.0 * cover_frac * (V - own_v) if world > 1 else 0.0,
            # sharded thinning: per round the keys (int64) after count, mask + dirty marks after select
            "thinning_zones_point_to_point": float(thin_rounds) * nb * zone * (8.0 + 1.0 + 1.0),
            "pair_affinities_all_reduce": 4.0 * pairs_per_voxel * V * 2.0 * (world - 1) / max(world, 1) if world > 1 else 0.0,
            "labels_all_reduce": 4.0 * thin_frac * V * 2.0 * (world - 1) / max(world, 1) if world > 1 else 0.0,
            "instances_all_gather": 4.0 * (V - own_v) if (result_gather and world > 1) else 0.0}
        out["ranks_plan"].append({
            "rank": r, "own_z": [oz0, oz1], "held_z": [lo, hi], "prediction_resident_gb": round(resident / 1e9, 2),
            "free_for_consensus_gb": round(budget / 1e9, 2), "tiles": [n, ny, nx], "cons_cache": bool(cache),
            "ring_z": int(ring[3]) if ring is not None else 0, "s1_work_per_owned_voxel": round(work, 3),
            "row_bytes_per_voxel": 4 * W,
            "received_gb_per_step": {k: round(v / 1e9, 3) for k, v in recv.items()},
            "received_gb_per_step_total": round(sum(recv.values()) / 1e9, 3)})
    out["replicated"] = {"watershed_host_loop_rank0_edges": int(0.21 * V), "pair_enumeration_patches": int(thin_frac * V)}
    return out


def to_instance_seg_tiled(pred_affs, foreground, mask_to_cover, numinst, patchshape, n_slabs,
                          **kw):
    """Single-process tiling: the whole prediction is resident, the consensus lives for one
    z-slab (or y/x tile of it, ``_yx_tiles``) at a time."""
    pred = backend.to_device_pred(pred_affs)
    shape = tuple(int(s) for s in pred.shape[1:])
    slabs = plan_slabs(shape[0], n_slabs)
    return assemble(pred, 0, shape, foreground, mask_to_cover, numinst, patchshape, slabs, **kw)


def _want_stream(pred_file, patchshape, kw):
    """Stream the prediction from the zarr store instead of loading it?  ``stream_prediction`` =
    True / False decides; "auto" (default, also PPP_STREAM_PRED=1 / 0): when the float16 array
    would take more than half of the free HBM."""
    import torch
    want = kw.pop("stream_prediction", os.environ.get("PPP_STREAM_PRED", "auto"))
    if want in (True, "1", 1):
        return True
    if want in (False, "0", 0):
        return False
    from .vote_instances import io_hdflike
    with io_hdflike.open_container(pred_file, "r") as f:
        key = kw.get("aff_key") or "volumes/pred_affs"
        if key not in f:
            return False
        nbytes = 2.0 * float(np.prod(f[key].shape))
    return torch.cuda.is_available() and nbytes > 0.5 * torch.cuda.mem_get_info()[0]


def _stitch_streamed(provider, foreground, numinst, bb, shape, patchshape, pred_file, result_folder, kw):
    """stitch_main with the prediction behind a provider: the bounding box becomes the volume the
    tiled assembly sees (a provider shifted by the box origin)."""
    import torch
    from . import postprocess
    from .vote_instances.vote_instances import write_result
    off = [int(b.start) for b in bb]
    bshape = tuple(int(b.stop - b.start) for b in bb)

    class Shifted:
        def pred_box(self, box):
            return provider.pred_box((box[0] + off[0], box[1] + off[0], box[2] + off[1], box[3] + off[1],
                                      box[4] + off[2], box[5] + off[2]))
    fg_bb = np.ascontiguousarray(foreground[bb])
    avail = torch.cuda.mem_get_info()[0]
    n, ny, nx = tiles_needed(bshape, patchshape, max(avail - 120.0 * float(np.prod(bshape)) - 8e9, 0.25 * avail),
                             safety=0.8 if getattr(provider, "expit", False) else 0.9, copies=2.0)
    kw = dict(kw, _instances_dtype=np.uint32, blockwise=False, return_intermediates=False)
    inst_bb, _ = assemble(Shifted(), 0, bshape, fg_bb, fg_bb.copy(), np.ascontiguousarray(numinst[bb]),
                          patchshape, plan_slabs(bshape[0], n), _yx_tiles=(ny, nx), **kw)
    instances = np.zeros(shape, dtype=np.uint32)
    instances[bb] = inst_bb
    if kw.get("remove_small_comps", 0) > 0:
        instances = postprocess.relabel(postprocess.remove_small_components(instances, kw["remove_small_comps"]))
    masked = instances.copy()
    masked[foreground == 0] = 0
    os.makedirs(result_folder, exist_ok=True)
    fn = os.path.splitext(os.path.basename(pred_file.rstrip("/")))[0]
    res_key = kw.get("res_key", "vote_instances")
    datasets = {res_key: instances.astype(np.uint16), "vote_foreground": foreground.astype(np.uint16),
                res_key + "_masked": masked.astype(np.uint16)}
    if kw.get("dilate_instances", False):
        dil = postprocess.dilate_instances(instances)
        datasets[res_key + "_dil_1"] = dil.astype(np.uint16)
        datasets[res_key + "_masked_dil_1"] = np.where(foreground == 0, 0, dil).astype(np.uint16)
    write_result(os.path.join(result_folder, fn + ".hdf"), datasets)
    return instances


def stitch_main(pred_file, result_folder=".", **kwargs):
    """Entry point behind ``vote_instances.stitch_patch_graph.main`` (reference
    stitch_patch_graph.py:672-894): bounding box of the cleaned foreground, assembly of the
    boxed volume (tiled when ``chunksize`` asks for it), result file with ``vote_instances``,
    ``vote_foreground`` and ``vote_instances_masked``."""
    import contextlib
    # (a streamed prediction keeps its container open while the tiles are read; it is closed
    # when the call ends, whatever way it ends)
    with contextlib.ExitStack() as stack:
        return _stitch_main(stack, pred_file, result_folder, **kwargs)


def _logits_in(arr, patchshape):
    """loadAffinities' test for logits, ``min < 0 and max > 1`` over the WHOLE array
    (utilVoteInstances.py:249-250), evaluated chunk by chunk along z so that a streamed
    prediction is never resident; stops as soon as both have been seen."""
    lo = hi = False
    step = max(1, int(getattr(arr, "chunks", arr.shape)[1]))
    for z0 in range(0, int(arr.shape[1]), step):
        blk = np.asarray(arr[:, z0:z0 + step])
        lo = lo or bool(blk.min() < 0)
        hi = hi or bool(blk.max() > 1)
        if lo and hi:
            return True
    return False


def _stitch_main(stack, pred_file, result_folder=".", **kwargs):
    import os
    from scipy import ndimage
    from .vote_instances import utilVoteInstances as util
    from .vote_instances.stitch_patch_graph import clean_mask
    from .vote_instances.vote_instances import write_result
    patchshape = np.array(kwargs["patchshape"])
    kw = dict(kwargs)
    kw.pop("patchshape")
    if pred_file.endswith(".npy"):
        # (C, Z, Y, X) array; foreground from the centre channel (utilVoteInstances.py:233-242,
        # whose own .npy branch only handles 2-d data)
        affinities = np.load(pred_file)
        if affinities.ndim == 3:
            affinities = affinities[:, None]
        mid = int(np.prod(patchshape)) // 2
        foreground = np.array(affinities[mid]) > util.getFgThreshold(**kw)
        numinst = 1 * foreground
    elif pred_file.rstrip("/").endswith(".zarr") and _want_stream(pred_file, patchshape, kw):
        # the prediction stays on disk: chunks are decoded on demand into a pinned host buffer and
        # copied to the device tile by tile (ZarrProvider); host memory holds the fields only
        from .vote_instances import io_hdflike
        f = stack.enter_context(io_hdflike.open_container(pred_file, "r"))
        aff_key = kw.setdefault("aff_key", "volumes/pred_affs")
        arr = f[aff_key]
        if len(arr.shape) != 4 or int(arr.shape[0]) != int(np.prod(patchshape)):
            raise NotImplementedError("streaming needs a channels-first (C, Z, Y, X) prediction array")
        for a in "zyx":
            if kw.get("crop_%s_s" % a, 0) or kw.get("crop_%s_e" % a) is not None:
                raise NotImplementedError("crops are not supported with a streamed prediction")
        numinst = util.maybeLoadNuminst(f, **kw)
        foreground, _ = util.loadFg(f, **dict(kw, patchshape=patchshape))
        # logits: loadAffinities' test over the whole array (utilVoteInstances.py:249-250), one
        # z-chunk at a time; `logits=True / False` in the configuration skips the scan
        logits = kw.get("logits")
        affinities = ZarrProvider(arr, expit=_logits_in(arr, patchshape) if logits is None else bool(logits))
    else:
        loaded = util.loadAffinities(pred_file, "", patchshape=patchshape, **kw)
        if loaded is None:
            return
        affinities, numinst, foreground = loaded
    foreground = np.squeeze(foreground)
    if foreground.ndim == 2:
        foreground = foreground[None]
    if numinst is None:
        numinst = foreground.astype(np.uint8)
    numinst = np.squeeze(numinst).reshape(foreground.shape)
    shape = foreground.shape
    # bounding box of the cleaned mask (stitch_patch_graph.py:745-767)
    mask = foreground
    if kw.get("ignore_small_comps", 0) > 0:
        # (stitch_patch_graph.py:749-751: np.ones([3] * ndim), i.e. 26-connectivity)
        mask = clean_mask(foreground, np.ones([3] * foreground.ndim), kw["ignore_small_comps"])
    if kw.get("only_bb", False) and mask.any():
        if kw.get("skeletonize_foreground"):
            # stitch_patch_graph.py:756-759: the bounding box is that of the SKELETON of the
            # cleaned mask (the cover mask itself is not thinned in blockwise mode,
            # vote_instances.py:219)
            from .vote_instances.vote_instances import _skeletonize
            mask = _skeletonize(mask, kw.get("skeletonize_backend"))
        nz = np.nonzero(mask)
        rad = patchshape // 2
        bb = tuple(slice(max(0, int(nz[i].min()) - int(rad[i])),
                         min(shape[i], int(nz[i].max()) + 1 + int(rad[i]))) for i in range(3))
    else:
        bb = tuple(slice(0, s) for s in shape)
    sub = (slice(None),) + bb
    n_slabs = 1
    if kw.get("chunksize") is not None and kw.get("blockwise", False):
        n_slabs = max(1, int(np.ceil((bb[0].stop - bb[0].start) / kw["chunksize"][0])))
    kw["blockwise"] = False
    # the reference's blockwise driver sets return_intermediates for its per-block calls
    # (default.toml:159, stitch_patch_graph.py:131-133); here the boxed volume is assembled as a
    # whole and the driver wants the instance map
    kw["return_intermediates"] = False
    if isinstance(affinities, ZarrProvider):
        return _stitch_streamed(affinities, foreground, numinst, bb, shape, patchshape, pred_file,
                                result_folder, kw)
    fg_bb = np.ascontiguousarray(foreground[bb])
    # the stitched volume carries uint32 ids (stitch_patch_graph.py:120): with the shipped
    # mws = true + includeSinglePatchCCS = true every selected patch is issued an id
    # (graph_mws.py:34-41), far more than 65 535 on a large volume
    kw["_instances_dtype"] = np.uint32
    inst_bb, _ = to_instance_seg_tiled(
        np.ascontiguousarray(affinities[sub]), fg_bb, fg_bb.copy(),
        np.ascontiguousarray(numinst[bb]), patchshape, n_slabs, **kw)
    instances = np.zeros(shape, dtype=np.uint32)
    instances[bb] = inst_bb
    # post-steps of the reference driver (stitch_patch_graph.py:831-894): small components
    # removed and the ids compacted -- on the uint32 map -- when remove_small_comps asks for it;
    # every dataset is then written as uint16 (the reference's astype: ids above 65 535 that
    # survive wrap, :852-870)
    from . import postprocess
    if kw.get("remove_small_comps", 0) > 0:
        instances = postprocess.relabel(
            postprocess.remove_small_components(instances, kw["remove_small_comps"]))
    if int(instances.max(initial=0)) > np.iinfo(np.uint16).max:
        logger.warning("instance ids up to %d are written as uint16 like the reference does "
                       "(stitch_patch_graph.py:852-856): set remove_small_comps > 0 to compact "
                       "them first", int(instances.max()))
    masked = instances.copy()
    masked[foreground == 0] = 0
    os.makedirs(result_folder, exist_ok=True)
    fn = os.path.splitext(os.path.basename(pred_file.rstrip("/")))[0]
    res_key = kw.get("res_key", "vote_instances")
    datasets = {res_key: instances.astype(np.uint16), "vote_foreground": foreground.astype(np.uint16),
                res_key + "_masked": masked.astype(np.uint16)}
    if kw.get("dilate_instances", False):
        dil = postprocess.dilate_instances(instances)
        datasets[res_key + "_dil_1"] = dil.astype(np.uint16)
        datasets[res_key + "_masked_dil_1"] = np.where(foreground == 0, 0, dil).astype(np.uint16)
    write_result(os.path.join(result_folder, fn + ".hdf"), datasets)
    return instances
