#!/usr/bin/env python
"""Benchmark of the MI355X-native stereo -> 8-view hot path.

  python bench.py --gpus N --steps K --warmup W

A step = one synthetic 1080p side-by-side frame per rank through the device-resident frame pipeline
(stm_d_adcensus_stm: cost init -> cross aggregation -> WTA -> DCC / IRV x5 / bilateral -> 6 DIBR views ->
interlacing).  Inputs are resident in HBM before the timed region.  Frames are independent, so ranks share no
data-path collective (scaling = weak); the only communication is the RCCL broadcast of the input batch from
rank 0 before timing starts.

N > 1: one process per GPU.  Under `python -m torch.distributed.run` (RANK / WORLD_SIZE in the environment) this
process is one of the ranks; started plainly (`python bench.py --gpus 8`) it launches torch.distributed.run itself as a
CHILD process -- before anything here touches the GPU -- and exits with the child's code.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured with a float4 copy)
AGG_KERNELS = ("pq_cost", "pq_h", "pq_v12", "pq_hw", "agg_h", "agg_v", "agg_hw", "cost_init")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--disp", type=int, default=64)
    ap.add_argument("--stages", type=int, default=3, help="1 = cost+agg+WTA (config 2), 2 = +refinement (config 3), 3 = full frame; add 256 for HSLO before WTA")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--agg-variant", type=int, default=0, help="0 = matrix-pipe aggregation (default), 10000 = vector-ALU kernels")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as a child job (nothing in THIS process
    has initialised the GPU) and hand its exit code back."""
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def cpu_baseline_and_parity(sbs, p, H, W, D, zd, stages, run_gpu):
    """The oracle (CPU restatement of the reference, kind 'port') timed on the host cores on a bounded sample of the same
    workload -- the whole frame when the host has the cores for it (about 3 s on 128 threads), otherwise the top
    quarter-height strip -- and, in the same run, the parity check SURVEY 8d asks for: the HIP pipeline is run on exactly
    that sample and every output is compared with the oracle's, element by element."""
    from oracle import pyoracle as orc
    orc.build()
    orc.limit_threads_to_usable_cpus()
    hslo = bool(stages & 0x100)

    def run(rows):
        part = np.ascontiguousarray(sbs[:rows])
        t0 = time.perf_counter()
        want = orc.adcensus_stm(part, rows, W, p.num_views, p.angle, D, zd, p.ad_coeff, p.census_coeff, p.ucd, p.lcd, p.usd,
                                p.lsd, p.thresh_s, p.thresh_h, hslo=hslo)
        return time.perf_counter() - t0, part, want

    rows = max(H // 4, 64) if orc.num_threads() < 8 else H
    rows = min(rows, H)
    dt, part, want = run(rows)
    fps = (1.0 / dt) * (rows / float(H))
    base = {"value": fps, "unit": "frames/s", "cores": orc.num_threads(), "kind": "port",
            "sample": "full pipeline (oracle/stm_oracle.c, OpenMP) on the top %dx%d rows of the same frame, D=%d: %.1f s wall x %d "
                      "threads; scaled by %d/%d to whole frames" % (W, rows, D, dt, orc.num_threads(), rows, H)}
    dl, dr, out = run_gpu(part, rows)
    s = stages & 0xff
    wl, wr = (want["wta_l"], want["wta_r"]) if s == 1 else (want["disp_l"], want["disp_r"])
    parity = {"disp_l_mismatch": int((dl != wl).sum()), "disp_r_mismatch": int((dr != wr).sum()),
              "interlaced_mismatch": int((out != want["interlaced"]).sum()) if s == 3 else None,
              "compared": "%dx%d rows of the benchmarked frame, HIP pipeline vs oracle, every element" % (W, rows)}
    return base, parity


def main():
    args = parse()
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world_env == 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))

    import torch
    import torch.distributed as dist
    import stm_amd
    from stm_amd import device_api as dev, sharding, synth

    rank = int(os.environ.get("RANK", "0"))
    world = world_env
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    stm_amd.lib()  # raises if the HIP library is missing: there is no fallback path
    torch.cuda.set_device(local_rank)
    rccl_world = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        rccl_world = dist.get_world_size()

    H, W, D = args.height, args.width, args.disp
    zd = D // 2
    p = dev.FrameParams(num_disp=D, zero_disp=zd)  # SURVEY 8d defaults: usd=34, lsd=17, 8 views, angle 18.43
    stm_amd.lib().stm_set_agg_variant(args.agg_variant)

    # ---- input batch: one frame per rank, generated on rank 0, broadcast over RCCL (xGMI) -------------
    batch = torch.zeros(world, H, 2 * W, 3, dtype=torch.uint8, device="cuda")
    sbs_host = None
    if rank == 0:
        frames = []
        for r in range(world):
            f, _ = synth.sbs_frame(H, W, D, zd, seed=synth.SEED + r)
            frames.append(f)
        sbs_host = frames[0]
        batch.copy_(torch.from_numpy(np.stack(frames)))
    sharding.broadcast_batch(batch, src=0)
    mine = sharding.shard_indices(world, rank, world)  # one frame per rank
    frame = batch[mine[0]].contiguous()

    dl = torch.zeros(H, W, dtype=torch.float32, device="cuda")
    dr = torch.zeros_like(dl)
    out = torch.zeros(H, W, 3, dtype=torch.uint8, device="cuda")

    def step():
        dev.d_adcensus_stm(frame, dl, dr, out, p, stages=args.stages)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n, profile):
        barrier()
        if profile:
            dev.prof_reset()
            dev.prof_enable(profile)  # HIP events on the launch stream, inside the timed region
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        barrier()
        dt = time.perf_counter() - t0
        if profile:
            dev.prof_enable(False)
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    for _ in range(args.warmup):
        step()
    # the timed region carries events around the aggregation kernels only (3-4 event pairs per frame: the roofline figures
    # come from them); a second, short run with events around every named kernel gives the per-kernel breakdown
    dt = timed(args.steps, 2)
    kern = {}
    if rank == 0:
        for name in AGG_KERNELS:
            n, ms = dev.prof_read(name)
            if n:
                kern[name] = {"launches": n, "avg_ms": ms / n, "total_ms": ms}
    nbreak = min(args.steps, 20)
    timed(nbreak, 1)
    other = {}
    if rank == 0:
        for name in AGG_KERNELS + ("cross_arms", "hslo_classes", "hslo_lr", "hslo_rl", "hslo_tb", "hslo_bt", "hslo_to_pq", "wta", "irv", "bilateral", "gaussian_max", "view_synth", "mux"):
            n, ms = dev.prof_read(name)
            if n:
                other[name] = ms / n
    # "wall-clock over >= 100 frames": when the driver asks for fewer steps, a second, un-profiled loop gives it
    n100 = max(100, args.steps)
    dt100 = timed(n100, False) if args.steps < 100 else dt

    if rank == 0:
        V = float(D) * H * W * 4
        HW = float(H) * W
        # algorithmic bytes per launch (compulsory inputs + outputs, each buffer once).  The matrix-pipe kernels
        # serve BOTH views per launch: pq_h = first horizontal pass; pq_v12 = both vertical passes fused (K2 of SURVEY 8d): V in, V out, 2 arm planes per view;
        # pq_hw = last pass + WTA: V in, 2 arm planes, disparity out.  Vector-ALU kernels (--agg-variant 10000): as round 1.
        hslo = bool(args.stages & 0x100)
        # By default the first pass computes the initial costs itself (no pq_cost launch): per view it reads four dword planes
        # (BGRX + census of both images) and two arm planes and writes V -- SURVEY 8d's K1; the frame then moves 8 V.
        fused_cost = "pq_cost" not in kern
        alg = {"pq_cost": 2 * V + 16 * HW, "pq_h": 2 * (V + 18 * HW) if fused_cost else 2 * (2 * V + 2 * HW),
               "pq_v12": 2 * (2 * V + 2 * HW), "pq_hw": 2 * (V + 6 * HW),
               "agg_h": (2 * V + 2 * HW) if hslo else 2 * (V + 2 * HW + 16 * HW), "agg_v": 2 * V + 2 * HW,
               "agg_hw": 2 * (V + 2 * HW + 4 * HW), "cost_init": 2 * V + 4 * 4 * HW}
        traffic_all, traffic_src = {}, None
        tj = os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")
        if os.path.exists(tj) and (H, W, D, args.agg_variant, args.stages) == (1080, 1920, 64, 0, 3):
            traffic_all = json.load(open(tj))
            traffic_src = "profiles/r02_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH x2 per MI355X_MICROARCH.md)"
        per_kernel = {}
        for k in AGG_KERNELS:
            if k in kern:
                ach = alg[k] / (kern[k]["avg_ms"] * 1e-3) / 1e9
                per_kernel[k] = {"achieved": ach, "frac": ach / HBM_PEAK_GBS, "avg_launch_ms": kern[k]["avg_ms"],
                                 "launches_per_frame": kern[k]["launches"] / float(args.steps), "algorithmic_bytes_per_launch": alg[k],
                                 "traffic": traffic_all.get(k, {}).get("traffic_bytes")}
        agg_names = [k for k in per_kernel if k not in ("pq_cost", "cost_init")]
        dom = max(agg_names, key=lambda k: kern[k]["total_ms"])
        stage_ms = sum(kern[k]["total_ms"] for k in per_kernel) / args.steps
        stage_bytes = sum(alg[k] * kern[k]["launches"] for k in per_kernel) / args.steps
        roofline = {"bound": "hbm", "kernel": "stm_k_" + dom, "achieved": per_kernel[dom]["achieved"], "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": per_kernel[dom]["frac"], "traffic": per_kernel[dom]["traffic"],
                    "traffic_source": traffic_src, "algorithmic_bytes_per_launch": alg[dom],
                    "avg_launch_ms": kern[dom]["avg_ms"], "kernels": per_kernel,
                    "agg_stage_ms_per_frame": stage_ms, "agg_stage_GBps": stage_bytes / (stage_ms * 1e-3) / 1e9,
                    "agg_stage_frac": stage_bytes / (stage_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        fps = world * args.steps / dt
        res = {
            "metric": "stereo->8-view frames/sec @1080p d=64; cost-agg HBM GB/s vs roofline",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%dx%d synthetic stereo frame, D=%d, zd=%d, %s, one frame per GPU per step" % (
                W, H, D, zd, {1: "cost init + cross aggregation + WTA (BASELINE config 2)",
                              2: "config 2 + DCC + IRV x5 + bilateral (config 3)",
                              3: "full stereo->8-view frame: cost init + cross aggregation + WTA + DCC/IRV x5/bilateral + 6 DIBR views + interlacing"}[args.stages & 0xff]
                + (" + scanline optimisation (HSLO) before WTA" if args.stages & 0x100 else "")),
                       "stages": args.stages, "usd": p.usd, "lsd": p.lsd, "views": p.num_views, "sharding": "frames, 1 per rank",
                       "aggregation": "matrix pipe (stm_kernels_aggm.hip)" if args.agg_variant == 0 else "agg_variant %d" % args.agg_variant},
            "rccl_world_size": rccl_world,
            "rate_over_100_frames": {"frames": n100, "frames_per_s": world * n100 / dt100, "ms_per_frame": dt100 / n100 * 1e3},
            "roofline": roofline,
            "kernels_ms": {k: round(v, 4) for k, v in other.items()},
            "kernels_ms_source": "a separate %d-frame run with HIP events around every named kernel (the timed region keeps events around the aggregation kernels only)" % nbreak,
        }
        if not args.no_cpu_baseline and world == 1:
            def run_gpu(part, rows):
                d_part = torch.from_numpy(part).cuda()
                a = torch.zeros(rows, W, dtype=torch.float32, device="cuda")
                b = torch.zeros_like(a)
                o = torch.zeros(rows, W, 3, dtype=torch.uint8, device="cuda")
                dev.d_adcensus_stm(d_part, a, b, o, p, stages=args.stages)
                torch.cuda.synchronize()
                return a.cpu().numpy(), b.cpu().numpy(), o.cpu().numpy()
            res["cpu_baseline"], res["parity"] = cpu_baseline_and_parity(sbs_host, p, H, W, D, zd, args.stages, run_gpu)
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
