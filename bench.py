#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X lifting-DWT path.

Metric (BASELINE.json): Gsamples/s of the 2-D forward float CDF 9/7, 8192x8192,
5 levels, device resident -- also quoted as a fraction of the 8 TB/s HBM3E peak via
the algorithmic bytes of SURVEY.md s8(d) (10.656 B per input sample).

A "step" is one pass of the hot path over one batch of `--images` distinct synthetic
8192^2 images (default 8) resident in HBM (out-of-place entry dwt_cdf97_2f_s2 semantics, one
kernel launch per level for the whole batch).  With N GPUs every rank transforms its
own batch (independent images: no data-path collective); value = all ranks' samples
divided by the slowest rank's time ("scaling": "weak").

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def algorithmic_bytes(w, h, levels, itemsize=4):
    """per level one read + one write of that level's input region."""
    total = 0
    for j in range(levels):
        total += 2 * itemsize * (-(-w // (1 << j))) * (-(-h // (1 << j)))
    return total


def cpu_baseline(size, levels):
    """libdwt's own CPU path on the host cores (rank 0, N=1 only): oracle/_ref when it
    was built (kind "reference"), else the bit-identical restatement (kind "port").
    Bounded sample: ONE size x size image transformed repeatedly for ~2 s per schedule, best
    single run, following dwt_util_perf_cdf97_2_s (src/libdwt.c:21444-21476; M=1)."""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # one GPU's share of the host (16 cores on the MI355X boxes); BENCH_CPU_THREADS overrides
    cores = int(os.environ.get("BENCH_CPU_THREADS", min(avail, 16)))
    os.environ["OMP_NUM_THREADS"] = str(cores)
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    import oraclelib

    kind = "reference"
    try:
        if not os.path.exists(oraclelib.REF_SO):
            raise FileNotFoundError
        lib = oraclelib.Reference()
        lib.lib.dwt_util_set_accel(0)
        lib.lib.dwt_util_set_num_workers(1)
        lib.lib.dwt_util_set_num_threads(cores)
    except Exception:
        kind = "port"
        lib = oraclelib.Oracle()
        lib.set_threads(cores)
    rng = np.random.default_rng(1234)
    # pitch as dwt_util_get_stride(.,2) would pick it (power-of-two pitches alias in cache)
    pitch_elems = size + 144 if size % 1024 == 0 else size
    buf = np.zeros((size, pitch_elems), np.float32)
    src = rng.random((size, size), dtype=np.float32)
    # the reference's plain loop (accel 0) and its "fast SSE" setting of examples/simple-perf
    # (accel 12, 4 workers); the port has one schedule.  Bounded: each runs for ~2 s of wall
    # time (at most 40 transforms), the best single transform counts (dwt_util_perf protocol).
    configs = [(0, 1), (12, 4)] if kind == "reference" else [(0, 1)]
    results = []
    for accel, workers in configs:
        if kind == "reference":
            lib.lib.dwt_util_set_accel(accel)
            lib.lib.dwt_util_set_num_workers(workers)
        best, runs, t_start = None, 0, time.perf_counter()
        while runs < 40 and (runs < 3 or time.perf_counter() - t_start < 2.0):
            buf[:, :size] = src
            t0 = time.perf_counter()
            lib.fwd("cdf97_2f_s", buf[:, :size], levels)
            dt = time.perf_counter() - t0
            if runs > 0:
                best = dt if best is None else min(best, dt)
            runs += 1
        results.append((size * size / best / 1e9, accel, workers, runs, best))
    results.sort(reverse=True)
    val, accel, workers, runs, best = results[0]
    detail = "; ".join(f"accel {a}/{w} workers: {v:.2f} Gsamples/s (best of {r - 1} runs, {b:.3f} s)" for v, a, w, r, b in results)
    return {"value": val, "unit": "Gsamples/s", "cores": cores, "kind": kind,
            "sample": f"1 image {size}x{size} float, {levels} levels, dwt_cdf97_2f_s, {cores} OpenMP threads, "
                      f"pitch {pitch_elems * 4} B; {detail}"}


def other_workload(args, dwt, torch, dist, world, rank, local_rank):
    """The other BASELINE.json configs through the same contract (one JSON line, whole-job rate,
    barrier + synchronize on both sides, max over ranks): config3 = int CDF 5/3 4096^2 3 levels
    forward + inverse; config4 = float 9/7 forward 4096^2 5 levels, 32 images per GPU (256 over 8
    GPUs); config5 = float 9/7 forward 3-D 1024^3 3 levels, out of place."""
    dev = torch.device("cuda", local_rank)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    w = args.workload
    if w == "config3":
        n, J, nb = 4096, 3, 16
        src = torch.randint(-32768, 32768, (nb, n, n), generator=gen, device=dev, dtype=torch.int32)
        dst = torch.empty_like(src)
        back = torch.empty_like(src)
        def step():
            dwt.transform2d_batch("cdf53_i", 0, src, dst, n * n * 4, nb, n * 4, n, n, J)
            dwt.transform2d_batch("cdf53_i", 1, dst, back, n * n * 4, nb, n * 4, n, n, J)
        units, unit, dtype = nb * n * n, "Gsamples/s", "i32"
        alg = 2 * algorithmic_bytes(n, n, J) * nb
        metric = "Gsamples/s CDF 5/3 2-D int forward+inverse, 4096^2 3-level"
        name = f"CDF 5/3 forward + inverse 2-D int32, {n}x{n}, {J} levels, {nb} device-resident images per step per GPU"
        check = lambda: bool(torch.equal(back, src))
    elif w == "config4":
        n, J, nb = 4096, 5, 32
        src = torch.rand((nb, n, n), generator=gen, device=dev, dtype=torch.float32)
        dst = torch.empty_like(src)
        def step():
            dwt.transform2d_batch("cdf97_s", 0, src, dst, n * n * 4, nb, n * 4, n, n, J)
        units, unit, dtype = nb * n * n, "Gsamples/s", "f32"
        alg = algorithmic_bytes(n, n, J) * nb
        metric = "Gsamples/s CDF 9/7 2-D fwd float, batch of 4096^2 5-level"
        name = f"CDF 9/7 forward 2-D float, {n}x{n}, {J} levels, {nb} device-resident images per step per GPU (256 over 8 GPUs)"
        check = lambda: True
    else:
        n, J = 1024, 3
        src = torch.rand((n, n, n), generator=gen, device=dev, dtype=torch.float32)
        dst = torch.empty_like(src)
        def step():
            dwt.transform3d_op(src, dst, n * 4, n * n * 4, n, n, n, J)
        units, unit, dtype = n ** 3, "Gvoxels/s", "f32"
        alg = sum(8 * ((n >> j) ** 3) for j in range(J))
        metric = "Gvoxels/s CDF 9/7 3-D fwd float, 1024^3 3-level"
        name = f"CDF 9/7 forward 3-D float, {n}^3, {J} levels, out of place (cdf97_3f_op semantics), one volume per step per GPU"
        check = lambda: True

    def barrier():
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    from libdwt_amd.batch import max_over_ranks
    elapsed = max_over_ranks(elapsed, device=dev)
    ok = check()
    if rank == 0:
        ach = alg * args.steps / elapsed / 1e9
        print(json.dumps({
            "metric": metric, "value": round(world * units * args.steps / elapsed / 1e9, 3), "unit": unit, "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": name, "parallelism": f"batch-sharded x{world}", "round_trip_exact": ok},
            "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                         "kernel": "whole step (all levels): algorithmic bytes / step time"},
        }))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=8192)
    ap.add_argument("--levels", type=int, default=5)
    ap.add_argument("--images", type=int, default=8, help="distinct images per step and per GPU")
    ap.add_argument("--inplace", action="store_true", help="time the in-place entry dwt_cdf97_2f_s instead of _s2")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--opt", action="append", default=[], help="backend option name=value (cpt, tile_pairs, waves, ...)")
    ap.add_argument("--workload", default="headline", choices=["headline", "config3", "config4", "config5"],
                    help="headline = BASELINE.json's metric (default); config3/4/5 = the other BASELINE configs, same JSON contract")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libdwt_amd has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    os.environ["DWT_HIP_DEVICE"] = str(local_rank)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL on ROCm, bound to this rank's GPU; used for the barrier and the max-reduce only
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import libdwt_amd as dwt

    dwt.dwt_util_init()
    for kv in args.opt:
        k, v = kv.split("=")
        dwt.set_option(k, int(v))
    stream = torch.cuda.current_stream()
    dwt.set_stream(stream.cuda_stream)

    if args.workload != "headline":
        other_workload(args, dwt, torch, dist if use_dist else None, world, rank, local_rank)
        if use_dist:
            dist.destroy_process_group()
        return

    n, J, nb = args.size, args.levels, args.images
    dev = torch.device("cuda", local_rank)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    src = torch.rand((nb, n, n), generator=gen, device=dev, dtype=torch.float32)
    dst = src.clone() if args.inplace else torch.empty_like(src)
    img_bytes = n * n * 4

    def step():
        if args.inplace:
            for k in range(nb):
                dwt.dwt_cdf97_2f_s(dst[k], n * 4, 4, n, n, n, n, J)
        else:
            dwt.transform2d_batch("cdf97_s", 0, src, dst, img_bytes, nb, n * 4, n, n, J)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    dwt.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    k_ms, k_launches = dwt.prof_read()
    dwt.prof_enable(False)

    from libdwt_amd.batch import max_over_ranks

    elapsed = max_over_ranks(elapsed, device=dev)

    samples = world * nb * n * n * args.steps
    value = samples / elapsed / 1e9
    alg = algorithmic_bytes(n, n, J)
    # dominant kernel: the level-0 sweep; per launch it reads and writes the whole
    # level-0 region of every image of the batch once
    images_per_launch = 1 if args.inplace else nb
    l0_bytes = 2 * 4 * n * n * images_per_launch
    l0_ms = k_ms / max(k_launches, 1)
    achieved = l0_bytes / (l0_ms * 1e-3) / 1e9 if k_launches else None

    # HBM traffic of the dominant kernel from the PMC passes of the committed profile
    # (profiles/<tag>_pmc_level0.json: FETCH_SIZE x2 [gfx950 correction] + WRITE_SIZE per
    # launch of 8 images); scaled to this run's images per launch.  None if absent.
    traffic, traffic_src = None, None
    try:
        import glob

        prof = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_level0.json")))[-1]
        pj = json.load(open(prof))
        if n == 8192 and "hbm_traffic_bytes_per_launch" in pj:
            traffic = pj["hbm_traffic_bytes_per_launch"] * l0_bytes / pj["algorithmic_bytes_per_launch"]
            traffic_src = os.path.relpath(prof, ROOT)
    except Exception:
        pass

    if rank == 0:
        out = {
            "metric": "Gsamples/s (= % HBM3E BW) CDF 9/7 2-D fwd float, 8192^2 5-level",
            "value": round(value, 3),
            "unit": "Gsamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"CDF 9/7 forward 2-D float, {n}x{n}, {J} levels, "
                                   f"{nb} device-resident images per step per GPU, "
                                   + ("in-place entry dwt_cdf97_2f_s" if args.inplace else "out-of-place entry (dwt_cdf97_2f_s2 semantics, batched)"),
                       "entry": "dwt_cdf97_2f_s" if args.inplace else "dwt_cdf97_2f_s2",
                       "images_per_step_per_gpu": nb, "parallelism": f"batch-sharded x{world}"},
            "hbm_frac_algorithmic": round(value * 1e9 * alg / (n * n) / (HBM_PEAK_GBS * 1e9) / world, 4),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1) if achieved else None, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None,
                         "traffic": round(traffic) if traffic else None, "traffic_source": traffic_src,
                         "kernel": "k_fwd_sweep<Cdf97S> level 0",
                         "bytes_per_launch": l0_bytes, "avg_launch_ms": round(l0_ms, 5), "launches": k_launches},
        }
        if world == 1 and not args.no_cpu:
            try:
                out["cpu_baseline"] = cpu_baseline(n, J)
            except Exception as e:  # the checker is optional equipment of the bench
                out["cpu_baseline"] = {"value": None, "unit": "Gsamples/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
