#!/usr/bin/env python3
"""Headline benchmark: Msamples/s of the per-pixel path-trace hot path on MI355X.

A "step" is one full render of the frame through the C ABI (ray generation ->
bounce loop -> tone-map; for N > 1 also the RCCL gather of the strips onto rank 0).
The scene is already resident in HBM when the timed region starts.

Workloads (the benchmark configurations c2..c5):
  c4  random-spheres scene (485 spheres), 3840x2160, 1000 spp, depth 50   [default, every N]
      the fixed 4K frame is split over the N GPUs in 8-row strips dealt round-robin
      (strong scaling): the same configuration at N = 1, 2, 4, 8.
  c3  random-spheres 1200x675, 500 spp, depth 50; also timed as a secondary figure of the
      default N = 1 line ("c3" object)
  c2  small scene 1920x1080, 100 spp, depth 50
  c5  10k-sphere grid scene 3840x2160, 256 spp, depth 50

Launch: python bench.py --gpus N.  For N > 1 without a launcher (no WORLD_SIZE in the environment) this
process starts the N ranks itself, as a child `python -m torch.distributed.run --nproc-per-node N bench.py ...`
before it touches a GPU, passes the child's output through and exits with its code; started under
torch.distributed.run it is one of the ranks.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3  # MI355X vector FP32 peak: 256 CU x 256 FLOP/clk x 2.4 GHz
FLOP_PER_TEST = 17        # one ray/sphere test as the reference states it: oc 3, oc.d 5, oc.oc - r^2 6, hb^2 - a*c 3
FLOP_PER_TEST_EXECUTED = 14  # what the kernel issues per test: 7 fused multiply-adds of the packed filter
FLOP_PER_RAY = 80         # per closest-hit query: d.d, hit point, normal, scatter / sky
PEAK_MFMA_F16_TFLOPS = 2516.6  # MI355X dense f16 matrix peak: 256 CU x 4 SIMD x 1024 FLOP/clk x 2.4 GHz
FLOP_PER_ROW_MFMA = 64    # the matrix-core filter: K = 32 multiply-adds per (ray, table row)

WORKLOADS = {
    "c2": dict(scene="small", width=1920, height=1080, spp=100, depth=50),
    "c3": dict(scene="random_spheres", width=1200, height=675, spp=500, depth=50),
    "c4": dict(scene="random_spheres", width=3840, height=2160, spp=1000, depth=50),
    "c5": dict(scene="grid10k", width=3840, height=2160, spp=256, depth=50),
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


def committed_summary(workload, args, world):
    """The newest committed rocprofv3 summary of `workload` at one GPU (profiles/rNN/<workload>_summary.json), or
    (None, None): counters cannot be read from inside this process, so the figures that come from them are taken
    from the profile of the same command."""
    if args.spp or world != 1:
        return None, None
    here = os.path.dirname(os.path.abspath(__file__))
    if not os.path.isdir(os.path.join(here, "profiles")):
        return None, None
    for r in sorted((d for d in os.listdir(os.path.join(here, "profiles")) if d.startswith("r")), reverse=True):
        f = os.path.join(here, "profiles", r, f"{workload}_summary.json")
        if os.path.exists(f):
            try:
                return json.load(open(f)), f"profiles/{r}/{workload}_summary.json"
            except ValueError:
                return None, None
    return None, None


def committed_busy(workload, args, world):
    """Shares of the kernel's shader cycles in which a SIMD's vector issue port (SQ_ACTIVE_INST_VALU x 4 cycles)
    and its matrix pipe (SQ_VALU_MFMA_BUSY_CYCLES) were busy, per SIMD (1024 of them), from the committed
    profile of the workload.  (None, None, None) without one."""
    d, src = committed_summary(workload, args, world)
    try:
        cycles = d["valu"]["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
        issue = 4.0 * d["valu"]["SQ_ACTIVE_INST_VALU"] / cycles
        mfma = d["mfma"]["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["mfma"]["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        return round(issue, 4), round(mfma, 4), src
    except (KeyError, TypeError, ZeroDivisionError):
        return None, None, None


def committed_traffic(workload, args, world):
    """HBM bytes per launch of render_kernel for `workload` at one GPU, from the separate rocprofv3
    --pmc passes whose summaries are committed under profiles/.  FETCH_SIZE is doubled as the gfx950 counter
    tallies 128-byte requests at 64; both counters report KiB.  None when no summary of this workload exists."""
    d, src = committed_summary(workload, args, world)
    try:
        return int((2 * d["fetch"]["FETCH_SIZE"] + d["write"]["WRITE_SIZE"]) * 1024), src
    except (KeyError, TypeError):
        return None, None


class Job:
    """One workload set up on this rank: scene resident, strip buffers and gather ready."""

    def __init__(self, name, args, rank, world, local_rank, dev, with_accum, ctx_env=None):
        import torch
        from raytracer_amd import Renderer, RenderParams, scenes
        from raytracer_amd.multi_gpu import StripGather

        self.torch = torch
        self.cfg = dict(WORKLOADS[name])
        if args.spp:
            self.cfg["spp"] = args.spp
        self.name, self.world, self.rank = name, world, rank
        self.W, self.H = self.cfg["width"], self.cfg["height"]
        self.scene = scenes.by_name(self.cfg["scene"])
        # tuning switches are read from the environment when a context is created
        saved = {k: os.environ.get(k) for k in (ctx_env or {})}
        os.environ.update(ctx_env or {})
        try:
            self.r = Renderer(local_rank)
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        self.r.set_scene(self.scene)
        self.p = RenderParams(self.W, self.H, samples_per_pixel=self.cfg["spp"], max_depth=self.cfg["depth"], seed=0,
                              strip_first=rank, strip_stride=world)
        self.gather = StripGather(self.W, self.H, rank, world, dev, stage_on_host=(args.backend == "gloo"))
        self.rgba = self.gather.local_buffer()
        # the first gather of a process group sets up its point-to-point connections: done here, on the empty
        # buffers, so that no timed step pays for it even with --warmup 0
        self.gather.gather(self.rgba)
        self.accum = torch.zeros((self.rgba.shape[0], self.W, 4), dtype=torch.float32, device=dev) if with_accum else None
        self.rays = torch.zeros(1, dtype=torch.int64, device=dev)
        self.stream = torch.cuda.current_stream().cuda_stream

    def step(self, seed, ev=None):
        self.p.seed = seed
        if ev:
            ev[0].record()
        self.r.render_device(self.p, self.rgba, self.accum, self.rays, self.stream)
        if ev:
            ev[1].record()
        frame = self.gather.gather(self.rgba)  # rank 0: the assembled H x W x 4 frame
        if ev:
            ev[2].record()
        return frame

    def run(self, steps, warmup, sync):
        """-> (wall seconds of `steps` steps, mean kernel ms, mean gather ms, rays of this rank)"""
        torch = self.torch
        for i in range(warmup):
            self.step(1000 + i)
        self.rays.zero_()
        ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(steps)]
        sync()
        t0 = time.perf_counter()
        for i in range(steps):
            self.step(i, ev[i])
        sync()
        dt = time.perf_counter() - t0
        kern = sum(a.elapsed_time(b) for a, b, _ in ev) / steps
        gath = sum(b.elapsed_time(c) for _, b, c in ev) / steps
        return dt, kern, gath, int(self.rays.item())

    def count_scans(self):
        """(scans, table blocks walked) of one frame at seed 0, from one more frame rendered through the host entry
        point outside any timed region.  A scan finds the closest sphere for the 64 rays of a wave; a block is 32
        table rows against those 64 rays (four matrix instructions).  A kernel that walks the whole table walks
        rows / 32 blocks per scan; the one that reads a long table only where a block's bounding sphere is met by
        some ray of the wave walks fewer (the blocks of its table of bounds count too)."""
        self.p.seed = 0
        self.r.render_host(self.p)
        return self.r.last_scans(), self.r.last_blocks()

    def roofline(self, kern_ms, rays_per_launch, args):
        n_s = self.scene.n_spheres
        achieved = rays_per_launch * (FLOP_PER_TEST * n_s + FLOP_PER_RAY) / (kern_ms * 1e-3) / 1e12
        executed = rays_per_launch * (FLOP_PER_TEST_EXECUTED * n_s + FLOP_PER_RAY) / (kern_ms * 1e-3) / 1e12
        traffic, traffic_src = committed_traffic(self.name, args, self.world)
        info = self.r.filter_describe()
        if info["in_use"]:
            # The sphere filter runs on the matrix cores (split-f16 bilinear form).  `achieved` prices the
            # products the kernel really issues against the dense f16 MFMA peak: scans x 64 lanes x table rows
            # x 64 FLOP (primary rays decided from their tile's list take no scan; a scan is issued for all 64
            # lanes of its wave).  What BOUNDS the kernel is named by `bound`: the busier of the SIMD's vector
            # issue port and its matrix pipe in the committed counter profile of this workload.
            rows = info.get("table_rows") or info["n_rows"]
            scans, blocks = self.count_scans()
            mfma = blocks * 32 * 64 * FLOP_PER_ROW_MFMA / (kern_ms * 1e-3) / 1e12
            issue_busy, mfma_busy, busy_src = committed_busy(self.name, args, self.world)
            bound = "mfma" if issue_busy is None or mfma_busy >= issue_busy else "valu_issue"
            return {
                "bound": bound,
                "achieved": round(mfma, 3),
                "peak": PEAK_MFMA_F16_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(mfma / PEAK_MFMA_F16_TFLOPS, 4),
                # shares of the kernel's cycles in which each SIMD's vector issue port / matrix pipe was busy
                # (rocprofv3 --pmc passes of this same command, committed; null at N > 1 or other sample counts)
                "issue_busy": issue_busy,
                "mfma_busy": mfma_busy,
                "busy_source": busy_src,
                "achieved_counts": "matrix products issued",
                "algorithmic_tflops": round(achieved, 3),
                "algorithmic_over_mfma_peak": round(achieved / PEAK_MFMA_F16_TFLOPS, 4),
                "fp32_valu_peak": PEAK_FP32_TFLOPS,
                "algorithmic_over_fp32_valu_peak": round(achieved / PEAK_FP32_TFLOPS, 4),
                "traffic": traffic,
                "traffic_source": traffic_src,
                "kernel": "render_kernel",
                "kernel_ms": round(kern_ms, 3),
                "flop_model": f"matrix: blocks walked*32 rows*64 lanes*{FLOP_PER_ROW_MFMA}; algorithmic: "
                              f"rays*({FLOP_PER_TEST}*{n_s}+{FLOP_PER_RAY}); rays/launch={rays_per_launch:.0f}, scans/launch={scans}, "
                              f"blocks/launch={blocks}, table rows={rows}",
                "scanned_ray_share": round(scans * 64 / max(rays_per_launch, 1), 4),
                # 1 when every scan walks the whole table; below 1 when a long table is read only where a block's
                # bounding sphere is met by some ray of the wave
                "table_share_walked": round(blocks * 32 / max(scans * rows, 1), 4),
                "note": "frac prices the matrix products issued (blocks walked x 32 rows x 64 lanes) against the matrix peak; with "
                        "bound = valu_issue the headroom is instruction count, not frac; brute_force is the same frame "
                        "with every ray scanned against every sphere",
            }
        return {
            "bound": "valu_fp32",
            "achieved": round(achieved, 3),
            "peak": PEAK_FP32_TFLOPS,
            "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_FP32_TFLOPS, 4),
            # the same launch priced at the 14 FLOP per test the packed filter really issues
            "executed_frac": round(executed / PEAK_FP32_TFLOPS, 4),
            "traffic": traffic,
            "traffic_source": traffic_src,
            "kernel": "render_kernel",
            "kernel_ms": round(kern_ms, 3),
            "flop_model": f"algorithmic rays*({FLOP_PER_TEST}*{n_s}+{FLOP_PER_RAY}); executed "
                          f"rays*({FLOP_PER_TEST_EXECUTED}*{n_s}+{FLOP_PER_RAY}); rays/launch={rays_per_launch:.0f}",
        }

    def brute_force(self, steps, sync):
        """The same frame with the primary-ray lists off (every ray takes a scan against every table row, the
        brute-force formulation of SceneManager::hitClosest), timed like the headline: what BASELINE's
        FLOP convention can be checked against."""
        info = self.r.filter_describe()
        if not info["in_use"]:
            return None
        rows = info.get("table_rows") or info["n_rows"]
        sdt, skern, _, srays = self.run(steps, 1, sync)
        scans, blocks = self.count_scans()
        n_s = self.scene.n_spheres
        products = blocks * 32 * 64 * FLOP_PER_ROW_MFMA
        alg = srays / steps * (FLOP_PER_TEST * n_s + FLOP_PER_RAY)
        return {
            "switch": "RT_AMD_PRIM_MIN=0 RT_AMD_BLOCK_CULL=0",
            "value": round(self.W * self.H * self.cfg["spp"] * steps / sdt / 1e6, 3),
            "unit": "Msamples/s",
            "ms_per_step": round(sdt / steps * 1e3, 3),
            "kernel_ms": round(skern, 3),
            "scans_per_launch": scans,
            "scanned_ray_share": round(scans * 64 / max(srays / steps, 1), 4),
            "products_flop": products,
            "frac": round(products / (skern * 1e-3) / 1e12 / PEAK_MFMA_F16_TFLOPS, 4),
            "algorithmic_tflops": round(alg / (skern * 1e-3) / 1e12, 3),
            "algorithmic_over_fp32_valu_peak": round(alg / (skern * 1e-3) / 1e12 / PEAK_FP32_TFLOPS, 4),
        }

    def describe(self):
        c = self.cfg
        return (f"{self.name}: {c['scene']} scene ({self.scene.n_spheres} spheres) {self.W}x{self.H}, {c['spp']} spp, "
                f"depth {c['depth']}, samplesPerScatter 1, no lights")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (invalidates the headline)")
    ap.add_argument("--accum", action="store_true", help="also write the 16 B/px radiance sums")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the c3 figure and the lists-off (brute-force) figure of the default N = 1 line")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --share-gpu rehearse the N > 1 path on a one-GPU box")
    ap.add_argument("--share-gpu", action="store_true", help="every rank uses GPU 0 (rehearsal only)")
    return ap.parse_args(argv)


def launch_ranks(args):
    """--gpus N > 1 called plainly: start the N ranks as a fresh child process (one per GPU, RCCL rendezvous on
    127.0.0.1) and hand its exit code back.  Nothing here has touched a GPU or imported torch: the ranks are
    started before any GPU call, and this process is never replaced."""
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.share_gpu:
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank} has no GPU of its own ({torch.cuda.device_count()} visible); "
                         f"--share-gpu rehearses N > 1 on one GPU")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    job = Job(args.workload, args, rank, world, local_rank, dev, args.accum)
    dt, kern_ms, gather_ms, my_rays = job.run(args.steps, args.warmup, sync)

    red_dev = dev if args.backend == "nccl" else torch.device("cpu")
    hi = torch.tensor([dt, kern_ms, gather_ms], dtype=torch.float64, device=red_dev)
    mine = torch.tensor([kern_ms, gather_ms], dtype=torch.float64, device=red_dev)
    rays_total = torch.tensor([my_rays], dtype=torch.int64, device=red_dev)
    every = [mine]
    if world > 1:
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays_total, op=dist.ReduceOp.SUM)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
    dt = float(hi[0])
    every = [[float(v[0]), float(v[1])] for v in every]

    # one more frame at seed 0, outside the timed region: its hash is the same for every N (the strip deal changes
    # no byte), so the N = 1, 2, 4, 8 lines can be checked against each other
    frame = job.step(0)
    sync()

    if rank == 0:
        cfg = job.cfg
        samples = job.W * job.H * cfg["spp"] * args.steps
        out = {
            "metric": "Msamples/sec (px*spp, depth<=50)",
            "value": round(samples / dt / 1e6, 3),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": job.describe(),
                "parallelism": f"{world} GPU(s), 8-row strips round-robin, RGBA8 gather to rank 0",
                "rays_per_sample": round(int(rays_total.item()) / samples, 4),
            },
            # per-rank means over the timed steps: the spread of kernel_ms is the load balance of the
            # strip deal, gather_ms the cost of assembling the frame on rank 0
            "per_rank": {
                "kernel_ms_min": round(min(k for k, _ in every), 3), "kernel_ms_max": round(max(k for k, _ in every), 3),
                "gather_ms_min": round(min(g for _, g in every), 3), "gather_ms_max": round(max(g for _, g in every), 3),
                "kernel_ms": [round(k, 3) for k, _ in every],
                "gather_ms": [round(g, 3) for _, g in every],
            },
            "frame_sha256_seed0": hashlib.sha256(frame.cpu().numpy().tobytes()).hexdigest()[:16],
            # dominant kernel: render_kernel on this rank; algorithmic FLOPs of one launch / its duration
            "roofline": job.roofline(kern_ms, my_rays / args.steps, args),
        }
        if world == 1 and args.workload in ("c4", "c5") and not args.spp and not args.no_secondary:
            bf = Job(args.workload, args, rank, world, local_rank, dev, args.accum, ctx_env={"RT_AMD_PRIM_MIN": "0", "RT_AMD_BLOCK_CULL": "0"})
            out["roofline"]["brute_force"] = bf.brute_force(max(min(args.steps, 3), 1), sync)
            del bf
        if world == 1 and args.workload == "c4" and not args.spp and not args.no_secondary:
            sec = Job("c3", args, rank, world, local_rank, dev, args.accum)
            n = max(args.steps, 5)
            sdt, skern, _, srays = sec.run(n, 1, sync)
            out["c3"] = {
                "workload": sec.describe(),
                "value": round(sec.W * sec.H * sec.cfg["spp"] * n / sdt / 1e6, 3),
                "unit": "Msamples/s",
                "ms_per_step": round(sdt / n * 1e3, 3),
                "roofline": sec.roofline(skern, srays / n, args),
            }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(job.scene, cfg)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
