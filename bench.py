#!/usr/bin/env python
"""Benchmark of the MI355X-native stereo -> 8-view hot path.

  python bench.py --gpus N --steps K --warmup W

A step = one synthetic 1080p side-by-side frame per rank through the device-resident frame pipeline
(stm_d_adcensus_stm: cost init -> cross aggregation -> WTA -> DCC / IRV x5 / bilateral -> 6 DIBR views ->
interlacing).  Inputs are resident in HBM before the timed region.  Frames are independent, so ranks share no
data-path collective (scaling = weak); the only communication is the RCCL broadcast of the input batch from
rank 0 before timing starts.  With N > 1 the pipelined scatter / gather loop of the C5 batch path
(sharding.FrameBatchPipeline) is timed beside it and reported as `batch_movement` (never `value`).

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
# f32 issue roof of the chip: 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz = 78.6 T adds/s.  On gfx950 the f32-input MFMAs run at
# exactly the f32 vector rate (MI355X_MICROARCH.md: 64 FLOP/clk/SIMD) ON the vector ALU, so this one number bounds the masked
# MFMA chains of the matrix-pipe kernels and every other vector instruction they issue.
ISSUE_PEAK_TADDS = 256 * 4 * 32 * 2.4e9 / 1e12
AGG_KERNELS = ("pq_cost", "pq_h", "pq_vtab", "pq_v12", "pq_hw", "agg_h", "agg_v", "agg_hw", "cost_init")
MATRIX_PIPE = ("pq_h", "pq_v12", "pq_hw")
GOLDEN = os.path.join(ROOT, "tests", "golden")


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
    ap.add_argument("--batch", type=int, default=0, help="--gpus N > 1, the `batch_movement` leg: frames per step (default N, one per rank); rank 0 "
                    "scatters every step's frames and gathers its outputs inside that leg's timed region, double-buffered (sharding.FrameBatchPipeline)")
    ap.add_argument("--no-batch-movement", action="store_true", help="--gpus N > 1: skip the `batch_movement` leg (`value` is always the resident-input form)")
    ap.add_argument("--no-extras", action="store_true", help="skip the two-frames-in-flight and real-content legs and the window statistics behind roofline.issue (profiling passes)")
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
    flags = "-O2 (checker build)"
    try:  # the timed leg runs the -O3 -march=native build of the same source, compiled on this host
        orc.select(orc.build_fast())
        flags = "-O3 -march=native -ffp-contract=off"
    except Exception as e:  # no compiler on the box: fall back to the checker build and say so
        flags += "; fast build failed: %r" % (e,)
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
            "sample": "full pipeline (oracle/stm_oracle.c, OpenMP, gcc %s) on the top %dx%d rows of the same frame, D=%d: %.1f s wall x %d "
                      "threads; scaled by %d/%d to whole frames" % (flags, W, rows, D, dt, orc.num_threads(), rows, H)}
    dl, dr, out = run_gpu(part, rows)
    s = stages & 0xff
    wl, wr = (want["wta_l"], want["wta_r"]) if s == 1 else (want["disp_l"], want["disp_r"])
    parity = {"disp_l_mismatch": int((dl != wl).sum()), "disp_r_mismatch": int((dr != wr).sum()),
              "interlaced_mismatch": int((out != want["interlaced"]).sum()) if s == 3 else None,
              "compared": "%dx%d rows of the benchmarked frame, HIP pipeline vs oracle, every element" % (W, rows)}
    return base, parity


def window_stats(L, R, p, H, W):
    """Sum of window lengths per direction and view, and the bytes of the vertical window table, from the arm planes the
    GPU itself builds (host API ca_cross on a one-plane volume: cross = UP, DOWN, LEFT, RIGHT)."""
    from stm_amd import host_api
    sum_h, sum_v, tab_bytes = 0, 0, 0
    for img in (L, R):
        cross, _ = host_api.ca_cross(img, np.zeros((1, H, W), np.float32), p.ucd, p.lcd, p.usd, p.lsd)
        up, down, left, right = [c.astype(np.int64) for c in cross]
        sum_h += int((left + right).sum())
        sum_v += int((up + down).sum())
        # stm_k_vwin_table: per tile of 16 rows x 4 columns a 32-byte header and 32 bytes per quad of the sweep
        ys = np.arange(H)[:, None]
        nn = up + down
        s0 = np.where(nn > 0, ys - up, 1 << 30)
        e0 = np.where(nn > 0, ys + down, -(1 << 30))
        Hp, Wp = (H + 15) // 16 * 16, (W + 3) // 4 * 4
        S = np.full((Hp, Wp), 1 << 30, np.int64)
        E = np.full((Hp, Wp), -(1 << 30), np.int64)
        S[:H, :W], E[:H, :W] = s0, e0
        S = S.reshape(Hp // 16, 16, Wp // 4, 4).min(axis=(1, 3))
        E = E.reshape(Hp // 16, 16, Wp // 4, 4).max(axis=(1, 3))
        nit = np.where(E > S, (E - (S & ~3) + 3) >> 2, 0)
        tab_bytes += int(32 * nit.size + 32 * nit.sum())
    return sum_h, sum_v, tab_bytes


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
    # STM_BENCH_REHEARSAL=gloo: the N-rank code path on a ONE-GPU box (RCCL refuses two ranks on one device): every rank uses cuda:0,
    # the process group is gloo, the batch moves through host tensors.  It checks that the multi-rank flow runs end to end; its
    # line is marked invalid and is not a measurement.
    rehearsal = os.environ.get("STM_BENCH_REHEARSAL", "") == "gloo" and world > 1
    torch.cuda.set_device(0 if rehearsal else local_rank)
    rccl_world = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        rccl_world = dist.get_world_size()

    H, W, D = args.height, args.width, args.disp
    zd = D // 2
    p = dev.FrameParams(num_disp=D, zero_disp=zd)  # SURVEY 8d defaults: usd=34, lsd=17, 8 views, angle 18.43
    stm_amd.lib().stm_set_agg_variant(args.agg_variant)
    # `value`: inputs resident on every rank before timing, one frame per rank and step.  The C5 batch path (scatter / gather per
    # step) is a second, separately timed leg for N > 1 (`batch_movement`): its first hardware run is the driver's, so it must
    # not be what the headline depends on.
    movement_leg = world > 1 and not args.no_batch_movement
    B = (args.batch if args.batch > 0 else world) if movement_leg else world
    per_rank = 1

    # ---- input batch: generated on rank 0 -------------------------------------------------------------
    move_dev = "cpu" if rehearsal else "cuda"  # where the scattered / gathered tensors of the movement leg live
    batch = torch.zeros(world, H, 2 * W, 3, dtype=torch.uint8, device="cuda")
    sbs_host = None
    if rank == 0:
        frames = [synth.sbs_frame(H, W, D, zd, seed=synth.SEED + r)[0] for r in range(world)]
        sbs_host = frames[0]
        batch.copy_(torch.from_numpy(np.stack(frames)))
    # inputs resident in HBM on every rank before the timed region: RCCL broadcast of the batch (north_star), one frame per rank
    if rehearsal:
        hb = batch.cpu()
        sharding.broadcast_batch(hb, src=0)
        batch.copy_(hb)
    else:
        sharding.broadcast_batch(batch, src=0)
    frame = batch[sharding.shard_indices(world, rank, world)[0]].contiguous()

    dl = torch.zeros(H, W, dtype=torch.float32, device="cuda")
    dr = torch.zeros_like(dl)
    out = torch.zeros(H, W, 3, dtype=torch.uint8, device="cuda")

    def steps(n):
        for _ in range(n):
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
        steps(n)
        barrier()
        dt = time.perf_counter() - t0
        if profile:
            dev.prof_enable(False)
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    steps(args.warmup)
    # the timed region carries events around the aggregation kernels only (four event pairs per frame: the roofline figures
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
    ALL_NAMES = AGG_KERNELS + ("cross_arms", "hslo_classes", "hslo_lr", "hslo_rl", "hslo_tb", "hslo_bt", "hslo_to_pq", "wta", "irv", "bilateral",
                               "gaussian_max", "view_synth", "mux", "synth_mux")

    def read_all():
        res = {}
        for name in ALL_NAMES:
            n, ms = dev.prof_read(name)
            if n:
                res[name] = ms / n
        return res
    other = read_all() if rank == 0 else {}
    # "wall-clock over >= 100 frames": when the driver asks for fewer steps, a second, un-profiled loop gives it
    n100 = max(100, args.steps)
    dt100 = timed(n100, False) if args.steps < 100 else dt

    movement = batch_movement_leg(torch, dist, dev, sharding, synth, args, p, H, W, D, zd, B, rank, world, move_dev, rehearsal,
                                  (dl, dr, out)) if movement_leg else None

    if rank == 0:
        V = float(D) * H * W * 4
        HW = float(H) * W
        frames_per_step = world  # whole job: one frame per rank and step
        my_frames = float(per_rank * args.steps)  # frames behind this rank's kernel records
        L_host, R_host = np.ascontiguousarray(sbs_host[:, :W]), np.ascontiguousarray(sbs_host[:, W:])
        if args.no_extras:  # profiling passes: no launches besides the timed pipeline (the per-kernel averages stay clean)
            sum_h = sum_v = tab_bytes = 0
        else:
            sum_h, sum_v, tab_bytes = window_stats(L_host, R_host, p, H, W)
        # algorithmic bytes per launch (compulsory inputs + outputs, each buffer once).  The matrix-pipe kernels
        # serve BOTH views per launch: pq_h = first horizontal pass, computing the initial costs itself (per view four dword
        # planes -- BGRX + census of both images -- and two arm planes in, V out: SURVEY 8d's K1); pq_vtab = the vertical
        # window table (four arm planes in, the table out); pq_v12 = both vertical passes fused (K2): V in, V out, 2 arm planes
        # per view (read through the table); pq_hw = last pass + WTA: V in, 2 arm planes, disparity out.  The frame moves 8 V.
        # Vector-ALU kernels (--agg-variant 10000): as round 1.
        hslo = bool(args.stages & 0x100)
        hsr = p.usd <= 36 and D <= 64 and args.agg_variant == 0 and not hslo  # the last pass + WTA runs stm_k_pq_hsr (stm_kernels_aggh.hip)
        fused_cost = "pq_cost" not in kern
        alg = {"pq_cost": 2 * V + 16 * HW, "pq_h": 2 * (V + 18 * HW) if fused_cost else 2 * (2 * V + 2 * HW),
               "pq_vtab": 4 * HW + tab_bytes, "pq_v12": 2 * (2 * V + 2 * HW),
               "pq_hw": 2 * (V + 6 * HW) + (2 * H * ((W + 15) // 16) * 256 if hsr else 0),  # (+ stm_k_pq_hsr's window table: 256 B per tile of 16 pixels)
               "agg_h": (2 * V + 2 * HW) if hslo else 2 * (V + 2 * HW + 16 * HW), "agg_v": 2 * V + 2 * HW,
               "agg_hw": 2 * (V + 2 * HW + 4 * HW), "cost_init": 2 * V + 4 * 4 * HW}
        # useful adds per launch = the reference's own count of float adds (one per window element, d_ca_cross_sum.cu:189-194,
        # 284-289): D x the sum of the window lengths over both views; pq_v12 does two vertical passes
        useful = {"pq_h": float(D) * sum_h, "pq_hw": float(D) * sum_h, "pq_v12": 2.0 * D * sum_v}
        traffic_all, traffic_src = {}, None
        tj = os.path.join(ROOT, "profiles", "r04_pmc_traffic.json")
        if os.path.exists(tj) and (H, W, D, args.agg_variant, args.stages) == (1080, 1920, 64, 0, 3):
            traffic_all = json.load(open(tj))
            traffic_src = ("profiles/r04_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of this command, FETCH x2 per "
                           "MI355X_MICROARCH.md; taken on the commit named in its 'commit' key (profiles/README.md)")
        per_kernel = {}
        for k in AGG_KERNELS:
            if k in kern:
                ach = alg[k] / (kern[k]["avg_ms"] * 1e-3) / 1e9
                per_kernel[k] = {"bound": "issue" if k in MATRIX_PIPE else "hbm",
                                 "achieved": ach, "frac": ach / HBM_PEAK_GBS, "avg_launch_ms": kern[k]["avg_ms"],
                                 "launches_per_frame": kern[k]["launches"] / my_frames, "algorithmic_bytes_per_launch": alg[k],
                                 "traffic": traffic_all.get(k, {}).get("traffic_bytes")}
                if k in useful and useful[k] > 0:
                    tadds = useful[k] / (kern[k]["avg_ms"] * 1e-3) / 1e12
                    per_kernel[k]["issue"] = {"useful_adds_per_launch": useful[k], "achieved_Tadds": tadds, "peak": ISSUE_PEAK_TADDS,
                                              "frac": tadds / ISSUE_PEAK_TADDS}
        agg_names = [k for k in per_kernel if k not in ("pq_cost", "cost_init", "pq_vtab")]
        dom = max(agg_names, key=lambda k: kern[k]["total_ms"])
        stage_ms = sum(kern[k]["total_ms"] for k in per_kernel) / my_frames
        stage_bytes = sum(alg[k] * kern[k]["launches"] for k in per_kernel) / my_frames
        roofline = {"bound": per_kernel[dom]["bound"], "kernel": {"pq_v12": "stm_k_pq_v12r" if p.usd <= 36 and args.agg_variant == 0 else "stm_k_pq_v12t", "pq_h": "stm_k_pq_hc", "pq_hw": "stm_k_pq_hsr" if hsr else "stm_k_pq_hs"}.get(dom, "stm_k_" + dom),
                    "achieved": per_kernel[dom]["achieved"], "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": per_kernel[dom]["frac"], "traffic": per_kernel[dom]["traffic"],
                    "traffic_source": traffic_src, "algorithmic_bytes_per_launch": alg[dom],
                    "avg_launch_ms": kern[dom]["avg_ms"],
                    "bound_note": "achieved / peak / frac are the HBM figures SURVEY 8d defines (algorithmic bytes / launch time vs 8 TB/s); the roof that "
                                  "binds the matrix-pipe kernels is the f32 issue rate: see 'issue' (useful adds = one per window element, as the reference "
                                  "counts them; peak = 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz)",
                    "issue": per_kernel[dom].get("issue"), "kernels": per_kernel,
                    "agg_stage_ms_per_frame": stage_ms, "agg_stage_GBps": stage_bytes / (stage_ms * 1e-3) / 1e9,
                    "agg_stage_frac": stage_bytes / (stage_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "window_sums": {"horizontal": sum_h, "vertical": sum_v, "mean_window_h": sum_h / (2 * HW), "mean_window_v": sum_v / (2 * HW)}}
        fps = frames_per_step * args.steps / dt
        res = {
            "metric": "stereo->8-view frames/sec @1080p d=64; cost-agg HBM GB/s vs roofline",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%dx%d synthetic stereo frame, D=%d, zd=%d, %s, %s" % (
                W, H, D, zd, {1: "cost init + cross aggregation + WTA (BASELINE config 2)",
                              2: "config 2 + DCC + IRV x5 + bilateral (config 3)",
                              3: "full stereo->8-view frame: cost init + cross aggregation + WTA + DCC/IRV x5/bilateral + 6 DIBR views + interlacing"}[args.stages & 0xff]
                + (" + scanline optimisation (HSLO) before WTA" if args.stages & 0x100 else ""),
                "one frame per GPU per step"),
                       "stages": args.stages, "usd": p.usd, "lsd": p.lsd, "views": p.num_views,
                       "sharding": "frames, 1 per rank and step, inputs resident before timing (RCCL broadcast of the batch from rank 0)",
                       "aggregation": "matrix pipe (stm_kernels_aggm.hip)" if args.agg_variant == 0 else "agg_variant %d" % args.agg_variant},
            "rccl_world_size": rccl_world,
            "rate_over_100_frames": {"frames": n100 * frames_per_step, "frames_per_s": frames_per_step * n100 / dt100, "ms_per_frame": dt100 / (n100 * frames_per_step) * 1e3},
            "roofline": roofline,
            "kernels_ms": {k: round(v, 4) for k, v in other.items()},
            "kernels_ms_source": "a separate %d-step run with HIP events around every named kernel (the timed region keeps events around the aggregation kernels only)" % nbreak,
        }
        if movement is not None:
            res["batch_movement"] = movement
        if world == 1 and not args.no_extras:
            res["rate_two_in_flight"] = two_in_flight(torch, dev, sbs_host, p, H, W, args.stages, max(30, min(args.steps, 100)))
            rc = real_content(torch, dev, synth, p, H, W, D, zd, args.stages, read_all)
            if rc:
                res["real_content"] = rc
            if (H, W, D, args.stages) == (1080, 1920, 64, 3):
                res["rate_incl_pcie"] = rate_incl_pcie(sbs_host, p, H, W, D, zd, 40)
                res["configs"] = config_sweep(torch, dev, synth)
        if rehearsal:
            res["invalid"] = "STM_BENCH_REHEARSAL=gloo: %d ranks on one GPU over gloo and host tensors -- a dry run of the multi-rank flow, not a measurement" % world
        bad = 0
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
            bad = sum(v for k, v in res["parity"].items() if k.endswith("_mismatch") and v)
            if bad:
                res["invalid"] = "the HIP pipeline differs from the oracle on the benchmarked frame (%d elements): 'value' is NOT a valid result" % bad
        print(json.dumps(res))
        if bad:
            sys.stdout.flush()
            sys.exit(1)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def batch_movement_leg(torch, dist, dev, sharding, synth, args, p, H, W, D, zd, B, rank, world, move_dev, rehearsal, bufs):
    """The C5 batch path as its own timed leg: per step a batch of B frames, rank 0 scatters every rank its frames and
    gathers the packed outputs (one record per frame: two disparity maps + the interlaced frame), the scatter of step k+1 and the
    gather of step k-1 in flight while step k computes.  Runs on every rank; returns the report on rank 0.  Never `value`."""
    dl, dr, out = bufs
    n = max(2, min(args.steps, 20))
    batch = None
    if rank == 0:
        batch = torch.from_numpy(np.stack([synth.sbs_frame(H, W, D, zd, seed=synth.SEED + (r % world))[0] for r in range(B)])).to(move_dev)
    pipe = sharding.FrameBatchPipeline(B, (H, 2 * W, 3), torch.uint8,
                                       {"disp_l": ((H, W), torch.float32), "disp_r": ((H, W), torch.float32), "interlaced": ((H, W, 3), torch.uint8)},
                                       move_dev, rank, world)

    def run_frame(fr, outs):
        if rehearsal:  # host tensors in and out (gloo): compute on the GPU through the resident buffers
            dev.d_adcensus_stm(fr.cuda(), dl, dr, out, p, stages=args.stages)
            outs["disp_l"].copy_(dl); outs["disp_r"].copy_(dr); outs["interlaced"].copy_(out)
            return
        dev.d_adcensus_stm(fr, outs["disp_l"], outs["disp_r"], outs["interlaced"], p, stages=args.stages)

    def sync():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
    pipe.run([batch] * 2 if rank == 0 else None, 2, run_frame)  # warm-up
    for k in pipe.stats:
        pipe.stats[k] = 0 if k == "batches" else 0.0
    sync()
    t0 = time.perf_counter()
    pipe.run([batch] * n if rank == 0 else None, n, run_frame)
    sync()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    per_rank = [None] * world
    dist.all_gather_object(per_rank, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in pipe.stats.items()})
    if rank != 0:
        return None
    dt = float(t.item())
    return {"frames_per_step": B, "steps": n, "frames_per_s": B * n / dt, "ms_per_step": dt / n * 1e3,
            "batch_bytes_in": pipe.bytes_in, "batch_bytes_out": pipe.bytes_out, "record_bytes_per_frame": pipe.rec_bytes,
            "per_rank_ms": per_rank,
            "how": "sharding.FrameBatchPipeline: one scatter + one gather (packed record) per step, every buffer allocated before the loop; "
                   "per_rank_ms = host milliseconds per phase summed over the steps (stage = rank 0 copying frames into the per-rank scatter "
                   "staging, wait_in / wait_out = blocked on the scatter / gather, compute_issue = launching the frames, assemble = rank 0 "
                   "putting the gathered records into frame order)"}


def rate_incl_pcie(sbs, p, H, W, D, zd, n):
    """SURVEY 8d metric 1, second figure: the host-buffer boundary, PCIe included -- frames written into the stream's pinned
    input buffer, results read in place (stm_stream_input_buffer / stm_stream_collect_view; what tools/host_rate.py (c) measures)."""
    from stm_amd import video
    fs = video.FrameStream(H, W, p)
    try:
        for _ in range(2):
            fs.input_buffer()[...] = sbs
            fs.submit_inplace()
        fs.collect_view()
        fs.collect_view()
        t0 = time.perf_counter()
        pending = 0
        for _ in range(n):
            if pending == 2:
                fs.collect_view()
                pending -= 1
            fs.submit_inplace()
            pending += 1
        while pending:
            fs.collect_view()
            pending -= 1
        dt = time.perf_counter() - t0
    finally:
        fs.close()
    return {"frames": n, "frames_per_s": n / dt, "ms_per_frame": dt / n * 1e3, "bytes_in_per_frame": int(sbs.nbytes),
            "bytes_out_per_frame": int(2 * H * W * 4 + H * W * 3),
            "how": "stm_stream_* with two frames in flight: H2D of frame k+1 || compute of frame k || D2H of frame k-1, pinned buffers written / read in place"}


def config_sweep(torch, dev, synth):
    """The other BASELINE configurations on this build, 10 frames each (parity-test cases, not bench lines: each is compared with the
    oracle at full size in tests/test_gpu_fullsize.py): frame rate, and for the aggregation kernel with the largest total time its
    average launch time and HBM fraction (algorithmic bytes as for the headline)."""
    out = {}
    for name, H, W, D, stages in (("C2_cost_agg_wta", 1080, 1920, 64, 1), ("C3_hslo_refine", 1080, 1920, 64, 259),
                                  ("C4_d128", 1080, 1920, 128, 3), ("C5_4k_d256", 2160, 3840, 256, 3)):
        try:
            zd = D // 2
            p = dev.FrameParams(num_disp=D, zero_disp=zd)
            sbs, _ = synth.sbs_frame(H, W, D, zd)
            d_sbs = torch.from_numpy(sbs).cuda()
            a = torch.zeros(H, W, dtype=torch.float32, device="cuda")
            b = torch.zeros_like(a)
            o = torch.zeros(H, W, 3, dtype=torch.uint8, device="cuda")
            for _ in range(2):
                dev.d_adcensus_stm(d_sbs, a, b, o, p, stages=stages)
            torch.cuda.synchronize()
            n = 10
            dev.prof_reset()
            dev.prof_enable(2)
            t0 = time.perf_counter()
            for _ in range(n):
                dev.d_adcensus_stm(d_sbs, a, b, o, p, stages=stages)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            dev.prof_enable(False)
            V, HW = float(D) * H * W * 4, float(H) * W
            alg = {"pq_h": 2 * (V + 18 * HW), "pq_v12": 2 * (2 * V + 2 * HW), "pq_hw": 2 * (V + 6 * HW)}
            ks = {}
            for k in alg:
                cnt, ms = dev.prof_read(k)
                if cnt:
                    ks[k] = {"avg_launch_ms": ms / cnt, "launches_per_frame": cnt / float(n),
                             "hbm_frac": alg[k] / (ms / cnt * 1e-3) / 1e9 / HBM_PEAK_GBS}
            dom = max(ks, key=lambda k: ks[k]["avg_launch_ms"] * ks[k]["launches_per_frame"]) if ks else None
            out[name] = {"size": [H, W, D], "stages": stages, "frames": n, "frames_per_s": n / dt, "ms_per_frame": dt / n * 1e3,
                         "dominant_kernel": dom, "kernels": ks}
            del d_sbs, a, b, o
        except Exception as e:  # a config that does not fit the box must not cost the headline its line
            out[name] = {"error": repr(e)}
    return out


def two_in_flight(torch, dev, sbs_host, p, H, W, stages, n):
    """Two frames in flight on one GPU: two host threads, each with its own HIP stream and workspace, feed the same frame
    pipeline (what stm_stream_* does per buffer slot).  Not `value`: that stays one frame per step."""
    import threading
    barrier = threading.Barrier(3)
    errs = []

    def worker():
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                d_sbs = torch.from_numpy(sbs_host).cuda()
                a = torch.zeros(H, W, dtype=torch.float32, device="cuda")
                b = torch.zeros_like(a)
                o = torch.zeros(H, W, 3, dtype=torch.uint8, device="cuda")
                for _ in range(3):
                    dev.d_adcensus_stm(d_sbs, a, b, o, p, stages=stages)
                st.synchronize()
                barrier.wait()
                for _ in range(n):
                    dev.d_adcensus_stm(d_sbs, a, b, o, p, stages=stages)
                st.synchronize()
        except Exception as e:  # pragma: no cover
            errs.append(repr(e))
            try:
                barrier.abort()
            except Exception:
                pass
    ths = [threading.Thread(target=worker) for _ in range(2)]
    for t in ths:
        t.start()
    try:
        barrier.wait()
    except threading.BrokenBarrierError:
        pass
    t0 = time.perf_counter()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    if errs:
        return {"error": errs[0]}
    return {"frames": 2 * n, "frames_per_s": 2 * n / dt, "ms_per_frame": dt / (2 * n) * 1e3,
            "how": "two host threads x own stream + workspace, %d frames each, same frame and parameters as 'value'" % n}


def real_content(torch, dev, synth, p, H, W, D, zd, stages, read_all):
    """The same pipeline on real image content: the reference's own img/bud_2 + bud_3 pair (committed as data fixtures under
    tests/golden/) tiled to the benchmark's size.  Compared with the oracle in tests/test_gpu_fullsize.py."""
    from stm_amd import bmp_io
    pl, pr = os.path.join(GOLDEN, "bud_2.bmp"), os.path.join(GOLDEN, "bud_3.bmp")
    if not (os.path.exists(pl) and os.path.exists(pr)):
        return None
    sbs = synth.tiled_sbs_frame(bmp_io.read_bmp(pl), bmp_io.read_bmp(pr), H, W)
    d_sbs = torch.from_numpy(sbs).cuda()
    a = torch.zeros(H, W, dtype=torch.float32, device="cuda")
    b = torch.zeros_like(a)
    o = torch.zeros(H, W, 3, dtype=torch.uint8, device="cuda")
    for _ in range(5):
        dev.d_adcensus_stm(d_sbs, a, b, o, p, stages=stages)
    torch.cuda.synchronize()
    n = 50
    t0 = time.perf_counter()
    for _ in range(n):
        dev.d_adcensus_stm(d_sbs, a, b, o, p, stages=stages)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dev.prof_reset()
    dev.prof_enable(1)
    for _ in range(10):
        dev.d_adcensus_stm(d_sbs, a, b, o, p, stages=stages)
    torch.cuda.synchronize()
    dev.prof_enable(False)
    return {"frames_per_s": n / dt, "ms_per_frame": dt / n * 1e3, "kernels_ms_real": {k: round(v, 4) for k, v in read_all().items()},
            "frame": "tests/golden/bud_2.bmp + bud_3.bmp (640x384, the reference's img/ pair) tiled to %dx%d (repeated in x, mirrored in y), D=%d, zd=%d" % (W, H, D, zd)}


if __name__ == "__main__":
    main()
