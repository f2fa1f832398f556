"""CPU tests of the host-side logic: BMP I/O, synthetic generator, frame sharding (gloo, world_size 2)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REF_IMG, ROOT


def test_bmp_roundtrip(stm, tmp_path):
    rng = np.random.RandomState(0)
    for (h, w) in [(5, 7), (4, 8), (3, 1)]:  # widths with and without row padding
        img = rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
        p = str(tmp_path / ("t%dx%d.bmp" % (h, w)))
        stm.bmp_io.write_bmp(p, img)
        assert np.array_equal(stm.bmp_io.read_bmp(p), img)


def test_bmp_c_twin_agrees(stm, tmp_path):
    from stm_amd import host_api
    img = np.random.RandomState(1).randint(0, 256, size=(9, 13, 3)).astype(np.uint8)
    p1, p2 = str(tmp_path / "a.bmp"), str(tmp_path / "b.bmp")
    stm.bmp_io.write_bmp(p1, img)
    host_api.bmp_write(p2, img)
    assert open(p1, "rb").read() == open(p2, "rb").read()
    assert np.array_equal(host_api.bmp_read(p1), img)


@pytest.mark.skipif(not os.path.isdir(REF_IMG), reason="reference img/ only exists in the build container")
def test_reference_images_load(stm):
    from stm_amd import host_api
    a = stm.bmp_io.read_bmp(os.path.join(REF_IMG, "bud_1.bmp"))  # 640x360 with 2 trailing bytes
    assert a.shape == (360, 640, 3)
    assert np.array_equal(a, host_api.bmp_read(os.path.join(REF_IMG, "bud_1.bmp")))
    f1 = stm.bmp_io.read_bmp(os.path.join(REF_IMG, "fish_1.bmp"))
    f2 = stm.bmp_io.read_bmp(os.path.join(REF_IMG, "fish_2.bmp"))
    assert f1.shape == (384, 640, 3) and np.array_equal(f1, f2)


def test_synth_is_deterministic_and_consistent():
    from stm_amd import synth
    L, R, off = synth.stereo_pair(60, 96, 16, 8)
    L2, R2, off2 = synth.stereo_pair(60, 96, 16, 8)
    assert np.array_equal(L, L2) and np.array_equal(R, R2) and np.array_equal(off, off2)
    assert off.min() >= -(8 - 1) + 2 and off.max() <= (16 - 8 - 1) - 2
    # background pixels that are not occluded reappear in the right image at x + off
    y, x = 2, 50
    if 0 <= x + off[y, x] < 96:
        assert (R[y] == L[y, x]).all(axis=1).any()
    sbs, _ = synth.sbs_frame(60, 96, 16, 8)
    assert sbs.shape == (60, 192, 3) and np.array_equal(sbs[:, :96], L) and np.array_equal(sbs[:, 96:], R)


def test_shard_indices_partition():
    from stm_amd import sharding
    for B in (1, 7, 8, 64):
        for G in (1, 2, 3, 8):
            parts = [sharding.shard_indices(B, r, G) for r in range(G)]
            assert sorted(sum(parts, [])) == list(range(B))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stm_amd import sharding
    B = 5
    batch = torch.zeros(B, 4, 6, 3, dtype=torch.uint8)
    if rank == 0:
        batch = (torch.arange(B * 4 * 6 * 3) % 251).to(torch.uint8).reshape(B, 4, 6, 3)
    sharding.broadcast_batch(batch, src=0)
    local = sharding.process_batch(batch, lambda f: f.to(torch.float32).sum(dim=(1, 2)), rank, world)
    full = sharding.gather_frames(local, B, rank, world, dst=0)
    if rank == 0:
        want = batch.to(torch.float32).sum(dim=(2, 3))
        q.put(bool(torch.equal(full, want)))
    dist.barrier()
    dist.destroy_process_group()


def _pipeline_worker(rank, world, port, q):
    """The C5 batch loop (FrameBatchPipeline): scatter of batch k+1 and gather of batch k-1 in flight while batch k computes."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stm_amd import sharding
    B, NB = 5, 4  # five frames per batch over two ranks (3 + 2), four batches
    batches = None
    if rank == 0:
        batches = [((torch.arange(B * 4 * 6 * 3) * (k + 3)) % 251).to(torch.uint8).reshape(B, 4, 6, 3) for k in range(NB)]
    pipe = sharding.FrameBatchPipeline(B, (4, 6, 3), torch.uint8, {"rowsum": ((4,), torch.float32), "neg": ((4, 6, 3), torch.uint8)},
                                       "cpu", rank, world)
    got = []
    calls = {"gather": 0, "scatter": 0}
    real_gather, real_scatter = dist.gather, dist.scatter

    def counting_gather(*a, **kw):
        calls["gather"] += 1
        return real_gather(*a, **kw)

    def counting_scatter(*a, **kw):
        calls["scatter"] += 1
        return real_scatter(*a, **kw)
    dist.gather, dist.scatter = counting_gather, counting_scatter
    stage_ids = [id(t) for slot in (pipe.stage or []) for t in slot]

    def run_frame(frame, outs):
        outs["rowsum"].copy_(frame.to(torch.float32).sum(dim=(1, 2)))
        outs["neg"].copy_(255 - frame)

    pipe.run(batches, NB, run_frame, on_result=lambda k, res: got.append((k, {n: t.clone() for n, t in res.items()})))
    dist.gather, dist.scatter = real_gather, real_scatter
    # one scatter and ONE gather per batch (the outputs of a frame travel as one packed record), every buffer made before the loop
    assert calls == {"gather": NB, "scatter": NB}, calls
    assert pipe.stats["batches"] == NB and stage_ids == [id(t) for slot in (pipe.stage or []) for t in slot]
    assert pipe.rec_bytes == 16 + 80 and pipe.bytes_out == B * pipe.rec_bytes and pipe.bytes_in == B * 4 * 6 * 3
    if rank == 0:
        assert pipe.stage is not None and len(pipe.stage[0]) == world and pipe.land is not None
        ok = [k for k, _ in got] == list(range(NB))
        for k, res in got:
            ok = ok and torch.equal(res["rowsum"], batches[k].to(torch.float32).sum(dim=(2, 3))) and torch.equal(res["neg"], 255 - batches[k])
        q.put(bool(ok))
    else:
        assert not got  # outputs are gathered to dst only
    dist.barrier()
    dist.destroy_process_group()


def test_frame_batch_pipeline_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_pipeline_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok


def test_frame_batch_pipeline_single_process():
    from stm_amd import sharding
    B, NB = 3, 3
    batches = [((torch.arange(B * 2 * 2) + k) % 7).to(torch.uint8).reshape(B, 2, 2) for k in range(NB)]
    pipe = sharding.FrameBatchPipeline(B, (2, 2), torch.uint8, {"twice": ((2, 2), torch.int32)}, "cpu", 0, 1)
    got = {}
    pipe.run(batches, NB, lambda f, o: o["twice"].copy_(f.to(torch.int32) * 2), on_result=lambda k, r: got.__setitem__(k, r["twice"].clone()))
    assert sorted(got) == [0, 1, 2] and all(torch.equal(got[k], batches[k].to(torch.int32) * 2) for k in range(NB))


def test_frame_sharding_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok


def test_bench_spawns_its_own_ranks_without_touching_the_gpu(monkeypatch):
    """`python bench.py --gpus 8` outside torch.distributed.run must start the 8 ranks as a CHILD job (one process per GPU,
    RCCL) before anything in the parent initialises the GPU, and hand the child's exit code back."""
    import importlib
    import sys
    sys.modules.pop("bench", None)
    bench = importlib.import_module("bench")
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "1"])
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    cuda_before = "torch" in sys.modules and sys.modules["torch"].cuda.is_initialized()
    try:
        bench.main()
        assert False, "main() must exit with the child's code"
    except SystemExit as e:
        assert e.code == 7
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "8", "--steps", "5", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
    if "torch" in sys.modules:
        assert sys.modules["torch"].cuda.is_initialized() == cuda_before  # the parent did not initialise the GPU
