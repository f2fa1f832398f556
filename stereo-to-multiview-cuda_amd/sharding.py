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
multi-GPU node by the builder (one-GPU boxes only) -- the driver's N-GPU run is the first hardware run.
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


class FrameBatchPipeline:
    """Double-buffered scatter -> compute -> gather over a sequence of frame batches.

    in_shape: shape of one input frame (e.g. (H, 2W, 3), uint8); out_specs: {name: (shape, dtype)} of the per-frame
    outputs.  run_frame(frame, outs) computes one frame into the dict of per-frame output tensors (views of the
    pipeline's buffers, so nothing is copied).  batches: on rank `src` a sequence of [B][...] tensors on `device`
    (ignored elsewhere).  on_result(k, {name: [B][...]}) is called on `dst` for every batch, in order."""

    def __init__(self, frames_per_batch, in_shape, in_dtype, out_specs, device, rank, world_size, src=0, dst=0):
        self.B, self.rank, self.world, self.src, self.dst = frames_per_batch, rank, world_size, src, dst
        self.n_max = (frames_per_batch + world_size - 1) // world_size
        self.mine = shard_indices(frames_per_batch, rank, world_size)
        self.inbuf = [torch.zeros((self.n_max,) + tuple(in_shape), dtype=in_dtype, device=device) for _ in range(2)]
        self.outbuf = [{k: torch.zeros((self.n_max,) + tuple(s), dtype=t, device=device) for k, (s, t) in out_specs.items()}
                       for _ in range(2)]

    def _scatter(self, k, batches):
        b = batches[k] if self.rank == self.src else None
        return scatter_frames(self.inbuf[k & 1], b, self.B, self.rank, self.world, self.src, async_op=True)

    def run(self, batches, n_batches, run_frame, on_result=None):
        if n_batches <= 0:
            return
        pending_in = self._scatter(0, batches)
        pending_out = [None, None]  # per buffer: list of (name, work, finish) of the gather still reading it

        def drain(slot, k):
            if pending_out[slot] is None:
                return
            res = {}
            for name, work, finish in pending_out[slot]:
                if work is not None:
                    work.wait()
                res[name] = finish()
            pending_out[slot] = None
            if on_result is not None and self.rank == self.dst:
                on_result(k, res)

        for k in range(n_batches):
            if pending_in is not None:
                pending_in.wait()  # this batch's frames have arrived
            pending_in = self._scatter(k + 1, batches) if k + 1 < n_batches else None  # in flight while batch k computes
            drain(k & 1, k - 2)  # the gather of batch k-2 must be done with this output buffer
            ins, outs = self.inbuf[k & 1], self.outbuf[k & 1]
            for i in range(len(self.mine)):
                run_frame(ins[i], {name: t[i] for name, t in outs.items()})
            pending_out[k & 1] = []
            for name, t in outs.items():
                work, finish = gather_frames(t, self.B, self.rank, self.world, self.dst, async_op=True)
                pending_out[k & 1].append((name, work, finish))
        for k in (n_batches - 2, n_batches - 1):
            if k >= 0:
                drain(k & 1, k)
