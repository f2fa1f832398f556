"""Frame-batch sharding across the GPUs of one node.

Frames are independent (adcensus_stm is a pure function of one side-by-side frame, d_io.cu:7-238), so the
path shards with no data-path collective: rank r of G processes frames r, r+G, r+2G, ...  The only
communication is moving the batch.  Two forms:

* `broadcast_batch` -- every rank receives the whole batch (north_star's wording; right when the batch is small
  and every rank wants it anyway);
* `FrameBatchPipeline` -- the C5 path of SURVEY 8e: rank `src` SCATTERS each rank its own frames (grouped
  send/recv underneath: RCCL over xGMI with the "nccl" backend, gloo on CPU for the tests), the per-frame
  outputs are GATHERED to `dst` only (not all-gathered: G times fewer bytes), and the loop is double-buffered:
  the scatter of batch k+1 and the gather of batch k-1 are in flight while batch k computes.

One process per GPU.  Status: exercised with world size 2 on gloo (tests/test_host_logic.py); never yet run on a
multi-GPU node by the builder (one-GPU boxes only) -- the driver's N-GPU run is the first hardware run.  bench.py's
headline for N > 1 is therefore the resident-input form (broadcast_batch before timing); the pipelined scatter / gather
loop is timed beside it as an extra (`batch_movement`).
"""
import torch
import torch.distributed as dist


def shard_indices(num_frames, rank, world_size):
    """Indices of the frames this rank processes (round-robin: balanced for any batch size)."""
    return list(range(rank, num_frames, world_size))


def broadcast_batch(batch, src=0):
    """Every rank receives the whole input batch [B][H][2W][3] from `src` (north_star: 'RCCL broadcast of
    the input batch over xGMI').  `batch` must be allocated with the right shape on every rank."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(batch, src=src)
    return batch


def _active():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def scatter_frames(local, batch, num_frames, rank, world_size, src=0, async_op=False):
    """Rank `src` holds `batch` [num_frames][...]; every rank receives ITS frames (shard_indices) into `local`
    [ceil(num_frames / G)][...] (shards shorter than that leave the tail untouched... it is zero-padded on the
    sender).  Returns the work handle when async_op (None for a single process)."""
    n_max = local.shape[0]
    if world_size == 1 or not _active():
        idx = shard_indices(num_frames, 0, 1)
        local[: len(idx)].copy_(batch[idx])
        return None
    scatter_list = None
    if rank == src:
        scatter_list = []
        for r in range(world_size):
            idx = shard_indices(num_frames, r, world_size)
            part = torch.zeros_like(local)
            if idx:
                part[: len(idx)].copy_(batch[idx])
            scatter_list.append(part)
        assert all(p.shape[0] == n_max for p in scatter_list)
    return dist.scatter(local, scatter_list, src=src, async_op=async_op)


def gather_frames(local, num_frames, rank, world_size, dst=0, async_op=False, out=None):
    """Collect per-frame outputs on `dst` ONLY.  `local` is [n_max][...] whose first len(shard_indices(...)) entries are
    this rank's results.  Synchronous form: returns [num_frames][...] on dst (None elsewhere).  async_op: returns
    (work, finish) where finish() -- called after work.wait() -- assembles and returns the same."""
    if world_size == 1 or not _active():
        res = local[: len(shard_indices(num_frames, 0, 1))]
        if out is not None:
            out.copy_(res)
            res = out
        return (None, (lambda: res)) if async_op else res
    n_max = (num_frames + world_size - 1) // world_size
    if local.shape[0] < n_max:  # a short shard: every rank must contribute the same shape
        pad = torch.zeros((n_max,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
        local = pad
    bufs = [torch.empty_like(local) for _ in range(world_size)] if rank == dst else None
    work = dist.gather(local, bufs, dst=dst, async_op=async_op)

    def finish():
        if rank != dst:
            return None
        res = out if out is not None else torch.empty((num_frames,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        for r in range(world_size):
            idx = shard_indices(num_frames, r, world_size)
            if idx:
                res[idx] = bufs[r][: len(idx)]
        return res
    return (work, finish) if async_op else finish()


def process_batch(batch, run_frame, rank, world_size):
    """Apply `run_frame(frame) -> tensor` to this rank's shard of `batch`; returns the stacked local results."""
    outs = [run_frame(batch[i]) for i in shard_indices(batch.shape[0], rank, world_size)]
    if not outs:
        return None
    return torch.stack(outs)


def pack_layout(out_specs, align=16):
    """Byte layout of one frame's outputs in ONE buffer, so that a batch needs a single gather: {name: (offset, nbytes, shape,
    dtype)} and the total size.  Fields are aligned to `align` bytes (a float32 view needs a multiple of 4)."""
    layout, off = {}, 0
    for name, (shape, dtype) in out_specs.items():
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        layout[name] = (off, nbytes, tuple(shape), dtype)
        off = (off + nbytes + align - 1) // align * align
    return layout, off


class FrameBatchPipeline:
    """Double-buffered scatter -> compute -> gather over a sequence of frame batches.

    in_shape: shape of one input frame (e.g. (H, 2W, 3), uint8); out_specs: {name: (shape, dtype)} of the per-frame
    outputs.  run_frame(frame, outs) computes one frame into the dict of per-frame output tensors (views into the
    pipeline's packed output buffer, so nothing is copied).  batches: on rank `src` a sequence of [B][...] tensors on `device`
    (ignored elsewhere).  on_result(k, {name: [B][...]}) is called on `dst` for every batch, in order; the tensors are views of
    a buffer that is reused two batches later.

    Round 4: every buffer of the loop exists before the first step (scatter staging per rank on `src`, gather landing and
    the assembled batch on `dst`: the timed loop allocates nothing), and the outputs of a frame are ONE packed byte record
    (pack_layout), so a step issues one scatter and one gather instead of one gather per output.  `stats` (host milliseconds
    per phase, summed over the batches of run()) tells a scaling run where its time went."""

    def __init__(self, frames_per_batch, in_shape, in_dtype, out_specs, device, rank, world_size, src=0, dst=0):
        self.B, self.rank, self.world, self.src, self.dst = frames_per_batch, rank, world_size, src, dst
        self.n_max = (frames_per_batch + world_size - 1) // world_size
        self.mine = shard_indices(frames_per_batch, rank, world_size)
        self.layout, self.rec_bytes = pack_layout(out_specs)
        self.inbuf = [torch.zeros((self.n_max,) + tuple(in_shape), dtype=in_dtype, device=device) for _ in range(2)]
        self.outbuf = [torch.zeros((self.n_max, self.rec_bytes), dtype=torch.uint8, device=device) for _ in range(2)]
        multi = world_size > 1 and _active()
        # rank src: one staging tensor per destination rank and buffer slot (a shard shorter than n_max keeps its zero tail)
        self.stage = [[torch.zeros_like(self.inbuf[0]) for _ in range(world_size)] for _ in range(2)] if (multi and rank == src) else None
        # rank dst: where the gather lands, and the batch in frame order
        self.land = [[torch.empty_like(self.outbuf[0]) for _ in range(world_size)] for _ in range(2)] if (multi and rank == dst) else None
        self.result = [torch.zeros((frames_per_batch, self.rec_bytes), dtype=torch.uint8, device=device) for _ in range(2)] if rank == dst else None
        self.shards = [shard_indices(frames_per_batch, r, world_size) for r in range(world_size)]
        self.bytes_in = frames_per_batch * self.inbuf[0][0].numel() * self.inbuf[0].element_size()
        self.bytes_out = frames_per_batch * self.rec_bytes
        self.stats = {"stage_ms": 0.0, "wait_in_ms": 0.0, "compute_issue_ms": 0.0, "wait_out_ms": 0.0, "assemble_ms": 0.0, "batches": 0}

    def frame_views(self, buf_row):
        """{name: tensor} views into one packed record (a row of the output buffer)."""
        return {name: buf_row[off:off + nb].view(dtype).view(shape) for name, (off, nb, shape, dtype) in self.layout.items()}

    def batch_views(self, packed):
        """{name: [B][...]} views into an assembled batch of packed records."""
        n = packed.shape[0]
        return {name: packed[:, off:off + nb].view(dtype).view((n,) + shape) for name, (off, nb, shape, dtype) in self.layout.items()}

    def _scatter(self, k, batches):
        import time
        slot = k & 1
        if self.world == 1 or not _active():
            idx = self.shards[0]
            self.inbuf[slot][: len(idx)].copy_(batches[k][idx])
            return None
        scatter_list = None
        if self.rank == self.src:
            t0 = time.perf_counter()
            b = batches[k]
            for r in range(self.world):
                idx = self.shards[r]
                if idx:
                    self.stage[slot][r][: len(idx)].copy_(b[idx])
            scatter_list = self.stage[slot]
            self.stats["stage_ms"] += (time.perf_counter() - t0) * 1e3
        return dist.scatter(self.inbuf[slot], scatter_list, src=self.src, async_op=True)

    def _gather(self, slot):
        if self.world == 1 or not _active():
            return None
        return dist.gather(self.outbuf[slot], self.land[slot] if self.rank == self.dst else None, dst=self.dst, async_op=True)

    def run(self, batches, n_batches, run_frame, on_result=None):
        import time
        if n_batches <= 0:
            return
        st = self.stats
        pending_in = self._scatter(0, batches)
        pending_out = [None, None]  # per buffer slot: (batch index, work) of the gather still reading it

        def drain(slot):
            if pending_out[slot] is None:
                return
            k, work = pending_out[slot]
            pending_out[slot] = None
            t0 = time.perf_counter()
            if work is not None:
                work.wait()
            t1 = time.perf_counter()
            st["wait_out_ms"] += (t1 - t0) * 1e3
            if self.rank != self.dst:
                return
            res = self.result[slot]
            if self.land is None:  # a single process: the local buffer is the batch
                res[: len(self.shards[0])].copy_(self.outbuf[slot][: len(self.shards[0])])
            else:
                for r in range(self.world):
                    idx = self.shards[r]
                    if idx:
                        res[idx] = self.land[slot][r][: len(idx)]
            st["assemble_ms"] += (time.perf_counter() - t1) * 1e3
            if on_result is not None:
                on_result(k, self.batch_views(res))

        for k in range(n_batches):
            slot = k & 1
            t0 = time.perf_counter()
            if pending_in is not None:
                pending_in.wait()  # this batch's frames have arrived
            st["wait_in_ms"] += (time.perf_counter() - t0) * 1e3
            pending_in = self._scatter(k + 1, batches) if k + 1 < n_batches else None  # in flight while batch k computes
            drain(slot)  # the gather of batch k-2 must be done with this output buffer
            t0 = time.perf_counter()
            ins, outs = self.inbuf[slot], self.outbuf[slot]
            for i in range(len(self.mine)):
                run_frame(ins[i], self.frame_views(outs[i]))
            st["compute_issue_ms"] += (time.perf_counter() - t0) * 1e3
            pending_out[slot] = (k, self._gather(slot))
            st["batches"] += 1
        for k in (n_batches - 2, n_batches - 1):
            if k >= 0:
                drain(k & 1)
