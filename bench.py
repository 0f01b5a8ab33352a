#!/usr/bin/env python3
"""Headline benchmark: Msamples/s of the per-pixel path-trace hot path on MI355X.

A "step" is one full render of the frame through the C ABI (ray generation ->
bounce loop -> tone-map; for N > 1 also the RCCL gather of the strips onto rank 0).
The scene is already resident in HBM when the timed region starts.

Workloads (the benchmark configurations c2..c5):
  c3  random-spheres scene (485 spheres), 1200x675, 500 spp, depth 50   [default]
      with N GPUs the frame grows to 1200 x (675*N) so that every GPU keeps the
      same amount of work (weak scaling); strips of 8 rows are dealt round-robin.
  c2  small scene 1920x1080, 100 spp, depth 50
  c4  random-spheres 3840x2160, 1000 spp, depth 50 (fixed frame split over the GPUs)
  c5  10k-sphere grid scene 3840x2160, 256 spp, depth 50 (fixed frame split over the GPUs)

Launch: python bench.py --gpus 1, or under torch.distributed.run for N > 1.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3  # MI355X vector FP32 peak: 256 CU x 256 FLOP/clk x 2.4 GHz
FLOP_PER_TEST = 17        # one ray/sphere test: oc 3, oc.d 5, oc.oc - r^2 6, hb^2 - a*c 3
FLOP_PER_RAY = 80         # per closest-hit query: d.d, hit point, normal, scatter / sky

WORKLOADS = {
    "c2": dict(scene="small", width=1920, height=1080, spp=100, depth=50, weak=False),
    "c3": dict(scene="random_spheres", width=1200, height=675, spp=500, depth=50, weak=True),
    "c4": dict(scene="random_spheres", width=3840, height=2160, spp=1000, depth=50, weak=False),
    "c5": dict(scene="grid10k", width=3840, height=2160, spp=256, depth=50, weak=False),
}


def cpu_baseline(scene, cfg, seconds_target=15.0):
    """Times the CPU oracle (a port of the reference algorithm, see oracle/rt_oracle.c) on a
    bounded sample of the same workload: the full frame at a reduced sample count."""
    from oracle.orc import Oracle, build

    build()
    orc = Oracle("f64")
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    W, H = cfg["width"], cfg["height"]
    t0 = time.perf_counter()
    orc.render(scene, W, H, 1, cfg["depth"], seed=0, want_accum=False)
    t1 = time.perf_counter() - t0
    spp = int(max(1, min(cfg["spp"], seconds_target / max(t1, 1e-3))))
    t0 = time.perf_counter()
    orc.render(scene, W, H, spp, cfg["depth"], seed=1, want_accum=False)
    dt = time.perf_counter() - t0
    return {
        "value": round(W * H * spp / dt / 1e6, 4),
        "unit": "Msamples/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{cfg['scene']} {W}x{H}, {spp} of {cfg['spp']} spp, depth {cfg['depth']}, {dt:.1f} s of CPU work",
    }


def committed_traffic(args, world):
    """HBM bytes per launch of render_kernel for the default command, from the separate rocprofv3 --pmc
    passes whose summary is committed under profiles/ (FETCH_SIZE and WRITE_SIZE cannot be read from
    inside this process).  FETCH_SIZE is doubled as the gfx950 counter tallies 128-byte requests at 64;
    both counters report KiB.  None for any other workload or GPU count."""
    if args.workload != "c3" or args.spp or world != 1:
        return None, None
    here = os.path.dirname(os.path.abspath(__file__))
    rounds = sorted(d for d in os.listdir(os.path.join(here, "profiles")) if d.startswith("r")) \
        if os.path.isdir(os.path.join(here, "profiles")) else []
    for r in reversed(rounds):
        f = os.path.join(here, "profiles", r, "pmc_summary.json")
        if os.path.exists(f):
            try:
                d = json.load(open(f))
                return int((2 * d["fetch"]["FETCH_SIZE"] + d["write"]["WRITE_SIZE"]) * 1024), f"profiles/{r}/pmc_summary.json"
            except (KeyError, ValueError):
                return None, None
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (invalidates the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --share-gpu rehearse the N > 1 path on a one-GPU box")
    ap.add_argument("--share-gpu", action="store_true", help="every rank uses GPU 0 (rehearsal only)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from raytracer_amd import Renderer, RenderParams, scenes
    from raytracer_amd.multi_gpu import StripGather

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with python -m torch.distributed.run --nproc-per-node N")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    cfg = dict(WORKLOADS[args.workload])
    if args.spp:
        cfg["spp"] = args.spp
    W = cfg["width"]
    H = cfg["height"] * (world if cfg["weak"] else 1)
    scene = scenes.by_name(cfg["scene"])

    r = Renderer(local_rank)
    r.set_scene(scene)
    p = RenderParams(W, H, samples_per_pixel=cfg["spp"], max_depth=cfg["depth"], seed=0,
                     strip_first=rank, strip_stride=world)
    gather = StripGather(W, H, rank, world, dev, stage_on_host=(args.backend == "gloo"))
    rgba = gather.local_buffer()
    accum = torch.zeros((rgba.shape[0], W, 4), dtype=torch.float32, device=dev)
    rays = torch.zeros(1, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step(seed):
        p.seed = seed
        r.render_device(p, rgba, accum, rays, stream)
        return gather.gather(rgba)  # rank 0: the assembled H x W x 4 frame

    for i in range(args.warmup):
        step(1000 + i)
    rays.zero_()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        p.seed = i
        ev[i][0].record()
        r.render_device(p, rgba, accum, rays, stream)
        ev[i][1].record()
        frame = gather.gather(rgba)
    sync()
    dt = time.perf_counter() - t0
    kernel_ms = [a.elapsed_time(b) for a, b in ev]

    red_dev = dev if args.backend == "nccl" else torch.device("cpu")
    t = torch.tensor([dt, sum(kernel_ms) / len(kernel_ms)], dtype=torch.float64, device=red_dev)
    rays_total = rays.clone().to(red_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays_total, op=dist.ReduceOp.SUM)
    dt = float(t[0])
    kern_ms = float(t[1])
    my_rays = int(rays.item())

    if rank == 0:
        samples = W * H * cfg["spp"] * args.steps
        n_s = scene.n_spheres
        # dominant kernel: render_kernel on this rank; algorithmic FLOPs of one launch / its duration
        rays_per_launch = my_rays / args.steps
        flops = rays_per_launch * (FLOP_PER_TEST * n_s + FLOP_PER_RAY)
        achieved = flops / (kern_ms * 1e-3) / 1e12
        traffic, traffic_src = committed_traffic(args, world)
        out = {
            "metric": "Msamples/sec (px*spp, depth<=50)",
            "value": round(samples / dt / 1e6, 3),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak" if cfg["weak"] else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: {cfg['scene']} scene ({n_s} spheres) {W}x{H}, {cfg['spp']} spp, "
                            f"depth {cfg['depth']}, samplesPerScatter 1, no lights",
                "parallelism": f"{world} GPU(s), 8-row strips round-robin, RGBA8 gather to rank 0",
                "rays_per_sample": round(int(rays_total.item()) / samples, 4),
            },
            "roofline": {
                "bound": "valu_fp32",
                "achieved": round(achieved, 3),
                "peak": PEAK_FP32_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_FP32_TFLOPS, 4),
                "traffic": traffic,
                "traffic_source": traffic_src,
                "kernel": "render_kernel",
                "kernel_ms": round(kern_ms, 3),
                "flop_model": f"rays*({FLOP_PER_TEST}*{n_s}+{FLOP_PER_RAY}), rays/launch={rays_per_launch:.0f}",
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, cfg)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
