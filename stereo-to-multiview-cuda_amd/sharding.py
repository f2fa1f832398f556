"""Frame-batch sharding across the GPUs of one node.

Frames are independent (adcensus_stm is a pure function of one side-by-side frame, d_io.cu:7-238), so the
path shards with no data-path collective: rank r of G processes frames r, r+G, r+2G, ...  The only
communication is moving the batch: rank 0 broadcasts the input frames (RCCL over xGMI when the backend is
"nccl"; gloo on CPU for tests) and the per-frame outputs are gathered back.  One process per GPU.
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


def gather_frames(local, num_frames, rank, world_size, dst=0):
    """Collect per-frame outputs on `dst`.  `local` is [n_local][...] for shard_indices(num_frames, rank, G);
    returns [num_frames][...] on dst (None elsewhere).  Shards are padded to equal length for all_gather."""
    if world_size == 1:
        return local
    n_max = (num_frames + world_size - 1) // world_size
    pad = torch.zeros((n_max,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world_size)]
    dist.all_gather(bufs, pad)
    if rank != dst:
        return None
    out = torch.empty((num_frames,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(world_size):
        idx = shard_indices(num_frames, r, world_size)
        out[idx] = bufs[r][: len(idx)]
    return out


def process_batch(batch, run_frame, rank, world_size):
    """Apply `run_frame(frame) -> tensor` to this rank's shard of `batch`; returns the stacked local results."""
    outs = [run_frame(batch[i]) for i in shard_indices(batch.shape[0], rank, world_size)]
    if not outs:
        return None
    return torch.stack(outs)
