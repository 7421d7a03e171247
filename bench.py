#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X lifting-DWT path.

Metric (BASELINE.json): Gsamples/s of the 2-D forward float CDF 9/7, 8192x8192,
5 levels, device resident -- also quoted as a fraction of the 8 TB/s HBM3E peak via
the algorithmic bytes of SURVEY.md s8(d) (10.656 B per input sample).

Workload (SURVEY.md s8e, strong scaling): a FIXED total batch of `--images` distinct
synthetic 8192^2 images (default 64) is sharded over the N GPUs, image b -> rank b*N//B;
a "step" is one pass of the hot path over the whole batch: every rank transforms its
resident shard (out-of-place entry, dwt_cdf97_2f_s2 semantics, one kernel launch per
level for the whole shard).  Images are independent: no data-path collective; RCCL does
the barrier and the max-over-ranks only.  value = total samples / slowest rank's time.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N ...        # starts the N rank processes itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

PLACEMENT.  The sweeps' rate depends on which physical memory backs source, destination and the
library's LL scratch relative to each other (profiles/r04_placement.md).  The library places
what it allocates (dwt_hip_alloc_batch: up to `--placements` candidates of destination + scratch, each
timed with the first two levels of the batch's own transform, untimed for the benchmark); the bench
allocates its shard through it, as a caller that keeps a batch resident would.  Beside `value` the line
carries `value_first_placement`: a short run (5 steps) on PLAIN first allocations with the library's
search off -- what a caller that ignores placement gets in this process.  `--placements 1` makes that
the whole run.

Prints ONE JSON line on rank 0.  For N > 1 the line also carries `batch_split`: the time to
scatter the whole batch from rank 0 and gather the coefficients back over RCCL (grouped
point-to-point, libdwt_amd.batch) -- reported beside the transform, never inside `value`.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
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


# ---------------------------------------------------------------------------------------
# N > 1 without a launcher: the parent starts the rank processes.  It makes NO GPU call
# (it never imports torch), and no process that touched the GPU is ever re-exec'ed.
# ---------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _log_dir():
    d = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        return d
    except OSError:
        import tempfile

        return tempfile.gettempdir()


def launch_ranks(n, argv, timeout_s=600.0, grace_s=10.0):
    """Start `n` rank processes of this script (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in their
    environment) and relay rank 0's JSON line.  The parent watches ALL children: when any of
    them exits non-zero the others get `grace_s` seconds and are then terminated (exactly the
    processes started here, by pid); after `timeout_s` everything still running is terminated.
    Whatever happened, the last complete JSON line rank 0 printed is relayed; the exit code is
    non-zero only when there is none.  Every rank's stderr (and stdout of ranks >= 1) goes to
    gpurun_out/bench_rank<r>.log."""
    import threading

    port = _free_port()
    logdir = _log_dir()
    procs, logs = [], []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        log = open(os.path.join(logdir, f"bench_rank{r}.log"), "w")
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else log, stderr=log, text=True))
    lines = []

    def drain():  # rank 0's stdout is the result; read it while the ranks run
        for ln in procs[0].stdout:
            if ln.startswith("{") and ln.rstrip().endswith("}"):
                lines.append(ln.rstrip())
            else:
                logs[0].write(ln)

    reader = threading.Thread(target=drain, daemon=True)
    reader.start()
    t_end = time.time() + timeout_s
    failed_at, why = None, None
    while True:
        codes = [p.poll() for p in procs]  # poll EVERY child (any() would stop at the first one running)
        if all(c is not None for c in codes):
            break
        now = time.time()
        if failed_at is None and any(c not in (None, 0) for c in codes):
            failed_at = now
            why = f"a rank exited with an error (exit codes so far {codes})"
        if now >= t_end:
            why = f"no completion within {timeout_s:.0f} s"
            break
        if failed_at is not None and now - failed_at >= grace_s:
            break
        time.sleep(0.2)
    for p in procs:  # whoever is still there: SIGTERM, then SIGKILL -- only the pids started above
        if p.poll() is None:
            p.terminate()
    t_kill = time.time() + 5
    for p in procs:
        try:
            p.wait(timeout=max(0.1, t_kill - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
    reader.join(timeout=5)
    for log in logs:
        log.close()
    codes = [p.returncode for p in procs]
    if lines:
        print(lines[-1], flush=True)
    if why or any(codes):
        sys.stderr.write(f"bench.py: {why or 'a rank failed'}; exit codes {codes}; per-rank logs in {logdir}/bench_rank*.log\n")
        for r in range(n):
            try:
                tail = open(os.path.join(logdir, f"bench_rank{r}.log")).read()[-1500:]
            except OSError:
                tail = ""
            if tail.strip():
                sys.stderr.write(f"--- rank {r} ---\n{tail}\n")
    return 0 if lines else next((c for c in codes if c and c > 0), 1)


# ---------------------------------------------------------------------------------------
# Nothing below may lose the line: side measurements run under a deadline, one line is printed
# ---------------------------------------------------------------------------------------
STATE = {"line": None, "printed": False, "transform_done": False}


def guarded(fn, timeout_s, device_index=None):
    """Run fn() on a daemon thread; ("ok", result) | ("error", text) | ("timeout", None).  A
    collective that never completes cannot be cancelled: the caller then prints what it has and
    leaves with os._exit, the thread is abandoned."""
    import threading

    box = {}

    def run():
        try:
            if device_index is not None:
                import torch

                torch.cuda.set_device(device_index)  # a new thread starts on device 0
            box["ok"] = fn()
        except BaseException as e:  # noqa: BLE001 -- reported, never raised into the bench
            box["error"] = f"{type(e).__name__}: {e}"

    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(timeout_s)
    if th.is_alive():
        return "timeout", None
    if "error" in box:
        return "error", box["error"]
    return "ok", box.get("ok")


def emit(out):
    """The ONE JSON line of the run (rank 0)."""
    if not STATE["printed"]:
        STATE["printed"] = True
        print(json.dumps(out), flush=True)


def start_deadline(rank, seconds):
    """Whole-run watchdog of a rank: at the deadline rank 0 prints the transform line if it has one
    (flagged), and the process leaves without waiting for anything."""
    import threading

    def fire():
        if rank == 0 and STATE["line"] is not None and not STATE["printed"]:
            line = dict(STATE["line"])
            line["watchdog"] = f"run deadline of {seconds:.0f} s reached; side measurements dropped"
            emit(line)
        sys.stdout.flush()
        os._exit(0 if (STATE["printed"] or (rank != 0 and STATE["transform_done"])) else 4)

    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


def finish(rank, out, split_fn, split_timeout, teardown, device_index=None):
    """`out` (rank 0) is the complete transform line.  The side measurement (the batch split over
    RCCL) runs under its own deadline; then rank 0 prints the line -- with the split's numbers, or
    with why there are none -- and the rank leaves."""
    STATE["transform_done"] = True
    if rank == 0:
        STATE["line"] = out
    hung = False
    if split_fn is not None:
        status, res = guarded(split_fn, split_timeout, device_index)
        if status == "ok":
            split = res
        elif status == "error":
            split = {"error": res}
        else:
            hung = True
            split = {"error": f"no completion within {split_timeout:.0f} s (point-to-point transfer hung); "
                              "the transform's numbers are unaffected"}
        if rank == 0 and split is not None:
            out["batch_split"] = split
    if rank == 0:
        emit(out)
    if hung:
        os._exit(0)  # a wedged collective cannot be torn down; the line is out
    status, _ = guarded(teardown, 30, device_index)
    if status != "ok":
        sys.stdout.flush()
        os._exit(0)


class Plane:
    """Barrier and max-over-ranks of the timing, on RCCL (device tensors) when it is healthy and
    on gloo (host tensors) otherwise -- the transform itself has no collective."""

    def __init__(self, torch, dist, dev, on_device):
        self.torch, self.dist, self.on_device = torch, dist, on_device
        self.dev = dev if on_device else torch.device("cpu")

    def barrier(self):
        if self.dist is None:
            return
        t = self.torch.zeros(1, dtype=self.torch.float32, device=self.dev)
        self.dist.all_reduce(t)
        t.item()

    def max(self, seconds):
        if self.dist is None:
            return float(seconds)
        t = self.torch.tensor([float(seconds)], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather_json(self, obj, limit=4096):
        """Every rank's small dict, on every rank, in rank order -- over HOST tensors (gloo in both group
        layouts), so that the per-rank part of the line never depends on RCCL.  A rank whose text exceeds
        `limit` bytes contributes {"truncated": true}."""
        if self.dist is None:
            return [obj]
        torch = self.torch
        raw = json.dumps(obj).encode()
        if len(raw) > limit:
            raw = b'{"truncated": true}'
        mine = torch.zeros(limit, dtype=torch.uint8)
        mine[:len(raw)] = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
        parts = [torch.zeros(limit, dtype=torch.uint8) for _ in range(self.dist.get_world_size())]
        self.dist.all_gather(parts, mine)
        out = []
        for t in parts:
            b = bytes(t.tolist()).rstrip(b"\0")
            try:
                out.append(json.loads(b.decode()))
            except ValueError:
                out.append({"unreadable": True})
        return out


def open_group(torch, dist, dev, dev_index, shared, pg_timeout):
    """One process per GPU: a default group with gloo for host tensors and RCCL ("nccl" on ROCm)
    for device tensors.  RCCL's first collective runs under a deadline on a side stream; if it
    fails or hangs the timing's barrier / max fall back to gloo and the line says so."""
    import datetime

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    to = datetime.timedelta(seconds=pg_timeout)
    if shared:
        dist.init_process_group("gloo", timeout=to)
        return Plane(torch, dist, dev, False), "gloo (ranks share a device)"
    dist.init_process_group("cpu:gloo,cuda:nccl", timeout=to)

    def first_collective():
        side = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(side):
            t = torch.ones(1, device=dev)
            dist.all_reduce(t)
            side.synchronize()
            return float(t.item())

    status, res = guarded(first_collective, min(pg_timeout, 150), dev_index)
    ok = status == "ok" and res == float(dist.get_world_size())
    # every rank must take the same branch: agree over gloo
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 1:
        return Plane(torch, dist, dev, True), "rccl"
    reason = res if status == "error" else ("first all_reduce hung" if status == "timeout" else "another rank failed")
    return Plane(torch, dist, dev, False), f"gloo (RCCL unusable: {reason})"


# ---------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1): libdwt's own CPU path, timed beside the GPU in the same run
# ---------------------------------------------------------------------------------------
def cpu_baseline(size, levels):
    """libdwt's own CPU path on the host cores: oracle/_ref when it was built (kind
    "reference"), else the bit-identical restatement (kind "port").  SURVEY.md s8(d):
    protocol of dwt_util_perf_cdf97_2_s (src/libdwt.c:21444-21476: minimum over N runs, M=1)
    on ONE size x size image, for (1 thread, all of this GPU's host cores) x (dense pitch,
    dwt_util_get_stride pitch) x (plain loop accel 0, the "fast SSE" setting accel 12 /
    4 workers of examples/simple-perf/simple.c:15-16).  Bounded: ~1.5 s or 12 runs per row, a single
    run where one takes more than 2 s (about 20 s for the eight rows on the GPU box's host)."""
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
    except Exception:
        kind = "port"
        lib = oraclelib.Oracle()
    rng = np.random.default_rng(1234)
    src = rng.random((size, size), dtype=np.float32)
    # dwt_util_get_stride(.,2) avoids power-of-two pitches (they alias in the CPU caches)
    pitches = [size, size + 144] if size % 1024 == 0 else [size]
    schedules = [(0, 1), (12, 4)] if kind == "reference" else [(0, 1)]
    # 1 thread and ALL PHYSICAL CORES of the node; and this GPU's share of them (16 on the 8-GPU node)
    phys = physical_cores()
    thread_counts = sorted({1, cores, phys})
    rows = []
    for pitch_elems in pitches:
        buf = np.zeros((size, pitch_elems), np.float32)
        for threads in thread_counts:
            for accel, workers in schedules:
                if threads == 1 and pitch_elems == size and size % 1024 == 0 and (accel, workers) != (0, 1):
                    continue  # one thread on the power-of-two pitch takes seconds per run (cache aliasing): one schedule is the sample
                if kind == "reference":
                    lib.lib.dwt_util_set_accel(accel)
                    lib.lib.dwt_util_set_num_workers(workers)
                    lib.lib.dwt_util_set_num_threads(threads)
                else:
                    lib.set_threads(threads)
                best, runs, timed, t_start = None, 0, 0, time.perf_counter()
                while runs < 13 and (timed < 2 or time.perf_counter() - t_start < 1.5):
                    buf[:, :size] = src
                    t0 = time.perf_counter()
                    lib.fwd("cdf97_2f_s", buf[:, :size], levels)
                    dt = time.perf_counter() - t0
                    runs += 1
                    # the first run warms caches and the OpenMP pool -- unless a run takes seconds (one
                    # thread on the power-of-two pitch): then it is the sample, the leg stays bounded
                    if runs > 1 or dt > 2.0:
                        best = dt if best is None else min(best, dt)
                        timed += 1
                    if dt > 2.0:
                        break
                rows.append({"threads": threads, "pitch_bytes": pitch_elems * 4, "accel": accel, "workers": workers,
                             "gsamples_per_s": round(size * size / best / 1e9, 3), "best_s": round(best, 4), "runs": timed})
    top = max(rows, key=lambda r: r["gsamples_per_s"])
    share = max((r for r in rows if r["threads"] == cores), key=lambda r: r["gsamples_per_s"])
    return {"value": top["gsamples_per_s"], "unit": "Gsamples/s", "cores": top["threads"], "kind": kind, "cpu_model": cpu_model(),
            "physical_cores": phys, "value_one_gpu_share": share["gsamples_per_s"], "cores_one_gpu_share": cores,
            "sample": f"1 image {size}x{size} float, {levels} levels, dwt_cdf97_2f_s in place, best single run per row "
                      f"(dwt_util_perf protocol, M=1); rows: threads in {{1, {cores} = one GPU's share of the host (min(CPUs this process may run "
                      f"on = {avail}, 16); BENCH_CPU_THREADS overrides), {phys} = all physical cores}} x pitch {{dense, dwt_util_get_stride}} x "
                      f"{{accel 0, accel 12 / 4 workers}}; value = best row: {top['threads']} OpenMP threads, pitch "
                      f"{top['pitch_bytes']} B, accel {top['accel']} / {top['workers']} workers",
            "rows": rows}


def physical_cores():
    """Physical cores among the CPUs this process may run on (distinct (package, core id) pairs of /proc/cpuinfo);
    falls back to the logical count."""
    try:
        allowed = os.sched_getaffinity(0)
    except AttributeError:
        allowed = set(range(os.cpu_count() or 1))
    cores, cpu, pkg = set(), None, None
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("processor"):
                cpu = int(ln.split(":")[1])
            elif ln.startswith("physical id"):
                pkg = int(ln.split(":")[1])
            elif ln.startswith("core id") and cpu in allowed:
                cores.add((pkg, int(ln.split(":")[1])))
    except (OSError, ValueError):
        pass
    return len(cores) or len(allowed)


def cpu_model():
    """Model string and logical CPU count of the host (BASELINE.md s4.5 asks for it beside every CPU figure)."""
    name = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                name = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return f"{name} ({os.cpu_count()} logical CPUs on the host)"


def _cpu_env():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
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
    except Exception:
        kind = "port"
        lib = oraclelib.Oracle()
    return lib, kind, cores


def cpu_baseline_config3(n, J):
    """BASELINE config 3 on the host cores: libdwt's dwt_cdf53_2f_i + dwt_cdf53_2i_i (src/libdwt.c:16304,
    18142) on ONE n x n int32 image, protocol of dwt_util_perf_cdf53_2_i (src/libdwt.c:21262: minimum over
    runs of forward + inverse, M = 1), rows = {1 thread, this GPU's host cores} x {dense pitch,
    dwt_util_get_stride pitch}; bounded to ~1.5 s or 8 runs per row (a single run where one takes > 2 s)."""
    import numpy as np

    lib, kind, cores = _cpu_env()
    rng = np.random.default_rng(7)
    src = rng.integers(-32768, 32768, size=(n, n), dtype=np.int32)
    pitches = [n, n + 144] if n % 1024 == 0 else [n]
    rows = []
    for pe in pitches:
        buf = np.zeros((n, pe), np.int32)
        for threads in sorted({1, cores}):
            if kind == "reference":
                lib.lib.dwt_util_set_accel(0)
                lib.lib.dwt_util_set_num_workers(1)
                lib.lib.dwt_util_set_num_threads(threads)
            else:
                lib.set_threads(threads)
            best, runs, timed, t_start = None, 0, 0, time.perf_counter()
            while runs < 9 and (timed < 2 or time.perf_counter() - t_start < 1.5):
                buf[:, :n] = src
                t0 = time.perf_counter()
                j = lib.fwd("cdf53_2f_i", buf[:, :n], J)
                lib.inv("cdf53_2i_i", buf[:, :n], j)
                dt = time.perf_counter() - t0
                runs += 1
                if runs > 1 or dt > 2.0:
                    best = dt if best is None else min(best, dt)
                    timed += 1
                if dt > 2.0:
                    break
            rows.append({"threads": threads, "pitch_bytes": pe * 4, "gsamples_per_s": round(n * n / best / 1e9, 4), "best_s": round(best, 4), "runs": timed})
    top = max(rows, key=lambda r: r["gsamples_per_s"])
    return {"value": top["gsamples_per_s"], "unit": "Gsamples/s", "cores": top["threads"], "kind": kind, "cpu_model": cpu_model(),
            "sample": f"1 image {n}x{n} int32, {J} levels, dwt_cdf53_2f_i + dwt_cdf53_2i_i in place (one forward + inverse pair counts "
                      f"n^2 samples, as in `value`), best run per row; value = best row: {top['threads']} OpenMP threads, pitch {top['pitch_bytes']} B",
            "rows": rows}


def cpu_baseline_config5(n):
    """BASELINE config 5 on the host: libdwt's cdf97_3f_op_sep_horizontal_s (src/volume-dwt.c:727; the schedule
    volume_perftest_fwd97op_s times, :2810-2881) on ONE n^3 float volume, ONE level -- the reference has no
    multi-level 3-D driver and no OpenMP in src/volume-dwt.c, so this is one thread whatever the host; minimum of
    two runs (the perf test's protocol: min over runs of secs per voxel)."""
    import ctypes as C

    import numpy as np

    lib, kind, _cores = _cpu_env()

    class Vol(C.Structure):
        _fields_ = [("size_x", C.c_int), ("size_y", C.c_int), ("size_z", C.c_int), ("stride_x", C.c_size_t),
                    ("stride_y", C.c_size_t), ("stride_z", C.c_size_t), ("data", C.c_void_p)]

    def vol(a):
        return Vol(a.shape[2], a.shape[1], a.shape[0], a.strides[2], a.strides[1], a.strides[0], a.ctypes.data)

    v = np.random.default_rng(1234).random((n, n, n), dtype=np.float32)
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        if kind == "reference":
            d = np.empty_like(v)
            lib.lib.cdf97_3f_op_sep_horizontal_s(C.byref(vol(v)), C.byref(vol(d)))
        else:
            d = v.copy()
            lib.vol("cdf97_3f_s", d)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return {"value": round(n ** 3 / best / 1e9, 5), "unit": "Gvoxels/s", "cores": 1, "kind": kind, "cpu_model": cpu_model(),
            "sample": f"1 volume {n}^3 float, ONE level, cdf97_3f_op_sep_horizontal_s out of place, best of 2 runs "
                      f"({best:.2f} s = {best / n ** 3 * 1e9:.2f} ns per voxel; single-threaded code in the reference)"}


def _profile_traffic(l0_bytes, n):
    """HBM bytes per level-0 launch from the committed PMC passes (profiles/*_pmc_level0.json:
    FETCH_SIZE x2 [gfx950 correction] + WRITE_SIZE), scaled to this run's bytes per launch.
    Counters cannot be read from inside the run, so this is NOT an in-run measurement."""
    try:
        import glob

        prof = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_level0.json")))[-1]
        pj = json.load(open(prof))
        if n == 8192 and "hbm_traffic_bytes_per_launch" in pj:
            return round(pj["hbm_traffic_bytes_per_launch"] * l0_bytes / pj["algorithmic_bytes_per_launch"]), os.path.relpath(prof, ROOT)
    except Exception:
        pass
    return None, None


def _event_times(torch, fn, reps, warm):
    """HIP-event time of each of `reps` calls of fn(i) on the current stream (which is the
    stream the library launches on); returns the per-call milliseconds."""
    for i in range(warm):
        fn(i)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for i, (a, b) in enumerate(evs):
        a.record()
        fn(i)
        b.record()
    torch.cuda.synchronize()
    return [a.elapsed_time(b) for a, b in evs]


def single_image_stats(torch, dwt, src, dst, n, J):
    """the libdwt.h entries on ONE device-resident image -- forward out of place
    (dwt_cdf97_2f_s2) and in place (dwt_cdf97_2f_s), inverse out of place (dwt_cdf97_2i_s2) and in
    place (dwt_cdf97_2i_s) -- HIP-event min and median over 100 calls after 20 warm-up calls, each
    call on a different image of the resident batch (so nothing is served from the 256 MiB Infinity
    Cache), against the same algorithmic bytes.  (Timing is data independent: the in-place legs run
    on whatever the previous leg left in the batch's output images.)"""
    nb = src.shape[0]
    alg = algorithmic_bytes(n, n, J)
    work = dst  # the batch's output images double as in-place work buffers

    def s2(i):
        k = i % nb
        dwt.dwt_cdf97_2f_s2(src[k], dst[k], n * 4, 4, n, n, n, n, J)

    def inplace(i):
        k = i % nb
        dwt.dwt_cdf97_2f_s(work[k], n * 4, 4, n, n, n, n, J)

    def inv_s2(i):
        k = i % nb
        dwt.dwt_cdf97_2i_s2(dst[k], src[k], n * 4, 4, n, n, n, n, J)  # coefficients -> (the image again)

    def inv_inplace(i):
        k = i % nb
        dwt.dwt_cdf97_2i_s(work[k], n * 4, 4, n, n, n, n, J)

    # the interleaved (in-place lifting) layout: dwt_cdf97_2f_inplace_s / dwt_cdf97_2i_inplace_s (src/libdwt.c:12926, 17474),
    # out of place (device-level entry) and in place (what the libdwt.h entries are); 2 x the image per call
    def il_fwd(i):
        k = i % nb
        dwt.transform2d_interleaved("cdf97_s", 0, 0, src[k], dst[k], n * 4, 4, n, n, n, n, J)

    def il_inv(i):
        k = i % nb
        dwt.transform2d_interleaved("cdf97_s", 1, 0, dst[k], src[k], n * 4, 4, n, n, n, n, J)

    def il_fwd_inplace(i):
        k = i % nb
        dwt.dwt_cdf97_2f_inplace_s(work[k], n * 4, 4, n, n, n, n, J)

    def il_inv_inplace(i):
        k = i % nb
        dwt.dwt_cdf97_2i_inplace_s(work[k], n * 4, 4, n, n, n, n, J)

    # (no dwt.tune here: the tuner leaves levels that fit the Infinity Cache -- all of one image -- to the launcher's rule)
    out = {"reps": 100, "warmup": 20, "algorithmic_bytes": alg}
    legs = [("s2", s2), ("inplace", inplace), ("inv_s2", inv_s2), ("inv_inplace", inv_inplace),
            ("il_fwd", il_fwd), ("il_inv", il_inv), ("il_fwd_inplace", il_fwd_inplace), ("il_inv_inplace", il_inv_inplace)]
    for name, fn in legs:
        if name in ("inplace", "il_fwd_inplace"):
            work.copy_(src)
        if name == "inv_s2":  # coefficients of the whole batch back in dst
            dwt.transform2d_batch("cdf97_s", 0, src, dst, n * n * 4, nb, n * 4, n, n, J)
        if name == "il_inv":  # every image of dst a valid interleaved transform
            for k in range(nb):
                il_fwd(k)
        ms = _event_times(torch, fn, 100, 20)
        mn, med = min(ms), statistics.median(ms)
        out[f"{name}_us_min"] = round(mn * 1e3, 1)
        out[f"{name}_us_median"] = round(med * 1e3, 1)
        # the interleaved entries against what they have to move: the image in and out once (2 x 4 B per sample)
        ref_bytes = 2 * 4 * n * n if name.startswith("il_") else alg
        out[f"frac_{name}"] = round(ref_bytes / (med * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    out["il_how"] = ("il_*: the interleaved layout's entries (dwt_cdf97_2f_inplace_s / _2i_inplace_s semantics), bit-exact default "
                     "path; frac_il_* = 2 x image bytes / median / 8 TB/s")
    return out


def extra_legs(torch, dwt, src, dst, n, J, nb, budget_s=6.0):
    """What SURVEY.md s8(d) asks for beside the headline, each a few calls under one small time budget:
    `batch_inverse` (the resident batch back: dwt_cdf97_2i_s2 semantics, one batched call per step) and `inplace_batch`
    (the in-place entry dwt_cdf97_2f_s image by image over the batch)."""
    import numpy as np

    t_end = time.perf_counter() + budget_s
    img_bytes = n * n * 4
    out = {}
    alg = algorithmic_bytes(n, n, J)

    def rate(ms, images):
        return {"ms_per_step": round(ms, 4), "gsamples_per_s": round(images * n * n / (ms * 1e-3) / 1e9, 2),
                "hbm_frac_algorithmic": round(images * alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}

    # batch inverse: coefficients (dst, as the headline left them) -> src
    dwt.tune("cdf97_s", 1, dst, src, img_bytes, nb, n * 4, n, n, J)
    ms = _event_times(torch, lambda i: dwt.transform2d_batch("cdf97_s", 1, dst, src, img_bytes, nb, n * 4, n, n, J), 5, 2)
    out["batch_inverse"] = dict(rate(statistics.median(ms), nb), images=nb, how="HIP events, median of 5 steps after dwt_hip_tune; one batched call per step")
    if time.perf_counter() < t_end:
        def inplace_step(i):
            for k in range(nb):
                dwt.dwt_cdf97_2f_s(dst[k], n * 4, 4, n, n, n, n, J)
        ms = _event_times(torch, inplace_step, 3, 1)
        out["inplace_batch"] = dict(rate(statistics.median(ms), nb), images=nb, how="dwt_cdf97_2f_s per image over the batch, median of 3 steps")
    return out


def host_pointer_leg(torch, dwt, n, J):
    """The drop-in call on an image in HOST memory (pageable, as a libdwt program has it), end to end incl. PCIe both ways.
    Measured at the START of the run: after the placement search's 100+ GiB arena has been mapped and returned, transfers
    in both directions at once run at about half their rate in this process (14.6 ms instead of 7.4: profiles/r06_notes.md s4)."""
    import numpy as np

    host = np.random.default_rng(1234).random((n, n), dtype=np.float32)
    ts = {"fwd": [], "inv": []}
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dwt.dwt_cdf97_2f_s(host, n * 4, 4, n, n, n, n, J)
        t1 = time.perf_counter()
        dwt.dwt_cdf97_2i_s(host, n * 4, 4, n, n, n, n, J)
        t2 = time.perf_counter()
        if rep:  # (the first call pins the image and allocates the staging)
            ts["fwd"].append((t1 - t0) * 1e3)
            ts["inv"].append((t2 - t1) * 1e3)
    return {"forward_ms": round(min(ts["fwd"]), 3), "inverse_ms": round(min(ts["inv"]), 3),
            "gsamples_per_s_forward": round(n * n / (min(ts["fwd"]) * 1e-3) / 1e9, 2),
            "how": "dwt_cdf97_2f_s / dwt_cdf97_2i_s on one image in pageable HOST memory, wall clock of the synchronous call, best of 3, "
                   "before anything else of the run: PCIe both ways included -- not comparable with `value`"}


def shard_sweep(torch, dwt, src, dst, n, J, nb):
    """Strong-scaling projection measured on ONE GPU (the transform has no collective: at N ranks every rank
    runs the same batched call on B/N images, so the N-rank step takes what a B/N-image call takes here):
    the step timed at B, B/2, B/4 and B/8 images per call on the resident, placed batch (HIP events,
    10 calls after 3 warm-up calls each), efficiency = rate(B/N) / rate(B), projected speed-up = N x that."""
    img_bytes = n * n * 4
    rows, rate = [], {}
    for div in (1, 2, 4, 8):
        k = nb // div
        if k < 1 or nb % div:
            continue
        dwt.tune("cdf97_s", 0, src[:k], dst[:k], img_bytes, k, n * 4, n, n, J)  # tile heights of this shard size (untimed)
        ms = _event_times(torch, lambda i: dwt.transform2d_batch("cdf97_s", 0, src[:k], dst[:k], img_bytes, k, n * 4, n, n, J), 10, 3)
        med = statistics.median(ms)
        rate[div] = k * n * n / (med * 1e-3) / 1e9
        rows.append({"ranks": div, "images_per_call": k, "ms_per_step": round(med, 4), "gsamples_per_s_per_gpu": round(rate[div], 2),
                     "efficiency": round(rate[div] / rate[1], 4)})
    proj = {str(d): round(d * rate[d] / rate[1], 3) for d in rate if d > 1}
    return rows, proj


def batch_split_times(torch, plane, src, total, n, rank, world, dev):
    """Scatter the whole batch from rank 0 and gather it back (libdwt_amd.batch, grouped
    point-to-point over RCCL/xGMI): the cost of a batch that starts and ends on one GPU."""
    from libdwt_amd import batch as B

    dist = plane.dist
    lo, hi = B.shard_range(total, rank, world)
    loop = world == 1  # --force-dist on one GPU: the block goes rank 0 -> rank 0 through RCCL's send / recv
    full = None
    if rank == 0:
        full = torch.empty((total, n, n), dtype=torch.float32, device=dev)
        full[lo:hi].copy_(src)
    res, intact = {}, None
    for rep in range(2):  # the first pass opens the RCCL point-to-point channels
        torch.cuda.synchronize()
        plane.barrier()
        t0 = time.perf_counter()
        local = B.scatter_images(full, total, (n, n), torch.float32, dev, loopback=loop)
        torch.cuda.synchronize()
        plane.barrier()
        t1 = time.perf_counter()
        back = B.gather_images(local, total, loopback=loop)
        if rep == 1 and rank == 0:
            intact = bool(torch.equal(back, full))
        torch.cuda.synchronize()
        plane.barrier()
        t2 = time.perf_counter()
        res = {"scatter_ms": (t1 - t0) * 1e3, "gather_ms": (t2 - t1) * 1e3}
        del local, back
    moved = (total - (hi - lo if rank == 0 and not loop else 0)) * n * n * 4
    t = torch.tensor([res["scatter_ms"], res["gather_ms"]], dtype=torch.float64, device=plane.dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    moved_t = torch.tensor([float(moved)], dtype=torch.float64, device=plane.dev)
    dist.broadcast(moved_t, 0)
    sc, ga = float(t[0]), float(t[1])
    b = float(moved_t[0])
    return {"scatter_ms": round(sc, 3), "gather_ms": round(ga, 3), "bytes_each_way": int(b),
            "scatter_GBps": round(b / sc / 1e6, 1), "gather_GBps": round(b / ga / 1e6, 1),
            "round_trip_intact": intact if rank == 0 else None,
            "how": ("ONE rank (--force-dist rehearsal): the batch rank 0 -> rank 0 and back through RCCL's grouped send / recv, no xGMI hop"
                    if loop else "rank 0 -> all ranks and back, grouped isend/irecv (batch_isend_irecv), root-egress bound")}


def seeded_images(torch, gen, out, first):
    """image k of the batch is seeded 1234 + k (its GLOBAL index, whatever rank holds it), generated on
    the device it lives on -- so a rank's shard is the same data at every N and any image can be regenerated alone."""
    for i in range(out.shape[0]):
        gen.manual_seed(1234 + first + i)
        torch.rand(out.shape[1:], generator=gen, out=out[i])
    return out


def raw_tensor(torch, dev, ptr, shape):
    """A torch view of device memory the library allocated (CUDA array interface)."""
    class _Dev:
        pass
    o = _Dev()
    o.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f4", "data": (int(ptr), False), "version": 3, "strides": None}
    return torch.as_tensor(o, device=dev)


def choose_placement(torch, dev, tries, make, run):
    """Up to `tries` allocations of a workload's buffers (`make()` -> tuple of tensors) behind spacers of
    different sizes, `run(bufs)` once to warm and three times timed (untimed for the benchmark), the
    fastest kept: the sweeps' rate depends on which physical memory backs their buffers (DESIGN s5)."""
    spacers_gib = [0, 7, 2.6, 50]
    cands, report = [], []
    for k in range(max(1, tries)):
        sp_gib = spacers_gib[k % len(spacers_gib)]
        free_b, _ = torch.cuda.mem_get_info(dev)
        spacer = None
        if k > 0:
            need = int(sp_gib * (1 << 30)) + cands[0][3] + (8 << 30)
            if free_b < need:
                break
            spacer = torch.empty(int(sp_gib * (1 << 30)), dtype=torch.uint8, device=dev) if sp_gib else None
        bufs = make()
        nbytes = sum(t.numel() * t.element_size() for t in bufs)
        run(bufs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            run(bufs)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        cands.append((dt, spacer, bufs, nbytes))
        report.append({"spacer_GiB": sp_gib, "ms_per_step": round(dt * 1e3, 4)})
    best = min(range(len(cands)), key=lambda i: cands[i][0])
    bufs = cands[best][2]
    keep_spacer = cands[best][1]  # stays allocated: it is what holds the placement
    cands = None
    torch.cuda.empty_cache()
    return bufs, keep_spacer, {"attempts": report, "chosen": best, "how": "untimed: 3 steps per candidate allocation of the buffers; DESIGN s5"}


def other_workload(args, dwt, torch, plane, world, rank, dev):
    """The other BASELINE.json configs through the same contract (one JSON line, whole-job rate,
    barrier + synchronize on both sides, max over ranks): config3 = int CDF 5/3 4096^2 3 levels
    forward + inverse, 16 images per GPU; config4 = float 9/7 forward 4096^2 5 levels, a fixed
    batch of 256 images sharded b*N//B (strong scaling); config5 = float 9/7 forward 3-D 1024^3
    3 levels, out of place, one volume per GPU.  Returns the line (rank 0) or None."""
    from libdwt_amd.batch import shard_range

    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    w = args.workload
    scaling = "weak"
    if w == "config3":
        n, J, nb = 4096, 3, 16
        def make():
            s_ = torch.randint(-32768, 32768, (nb, n, n), generator=gen, device=dev, dtype=torch.int32)
            return s_, torch.empty_like(s_), torch.empty_like(s_)
        def run(b):
            dwt.transform2d_batch("cdf53_i", 0, b[0], b[1], n * n * 4, nb, n * 4, n, n, J)
            dwt.transform2d_batch("cdf53_i", 1, b[1], b[2], n * n * 4, nb, n * 4, n, n, J)
        units, unit, dtype = world * nb * n * n, "Gsamples/s", "i32"
        alg = 2 * algorithmic_bytes(n, n, J) * nb
        metric = "Gsamples/s CDF 5/3 2-D int forward+inverse, 4096^2 3-level"
        name = f"CDF 5/3 forward + inverse 2-D int32, {n}x{n}, {J} levels, {nb} device-resident images per step per GPU"
        check = lambda b: {"round_trip_exact": bool(torch.equal(b[2], b[0]))}
    elif w == "config4":
        n, J, total = 4096, 5, 256
        lo, hi = shard_range(total, rank, world)
        nb = hi - lo
        def make():
            s_ = seeded_images(torch, gen, torch.empty((nb, n, n), device=dev, dtype=torch.float32), lo)
            return s_, torch.empty_like(s_)
        def run(b):
            dwt.transform2d_batch("cdf97_s", 0, b[0], b[1], n * n * 4, nb, n * 4, n, n, J)
        units, unit, dtype = total * n * n, "Gsamples/s", "f32"
        alg = algorithmic_bytes(n, n, J) * nb
        scaling = "strong"
        metric = "Gsamples/s CDF 9/7 2-D fwd float, batch of 256 x 4096^2 5-level"
        name = (f"CDF 9/7 forward 2-D float, {n}x{n}, {J} levels, fixed batch of {total} device-resident images "
                f"sharded b*N//B over {world} GPU(s) ({nb} on rank 0)")
        def check(b):
            # in-run sanity: the first and the last image of this rank's shard, transformed ALONE through
            # the libdwt.h entry dwt_cdf97_2f_s2, must give the batch's bits
            one = torch.empty((n, n), dtype=torch.float32, device=dev)
            same = True
            for k in sorted({0, nb - 1}):
                dwt.dwt_cdf97_2f_s2(b[0][k], one, n * 4, 4, n, n, n, n, J)
                torch.cuda.synchronize()
                same = same and bool(torch.equal(one.view(torch.int32), b[1][k].view(torch.int32)))
            return {"batch_equals_single_image_entry": same}
    else:
        n, J = 1024, 3
        def make():
            s_ = torch.rand((n, n, n), generator=gen, device=dev, dtype=torch.float32)
            return s_, torch.empty_like(s_)
        def run(b):
            dwt.transform3d_op(b[0], b[1], n * 4, n * n * 4, n, n, n, J)
        units, unit, dtype = world * n ** 3, "Gvoxels/s", "f32"
        alg = sum(8 * ((n >> j) ** 3) for j in range(J))
        metric = "Gvoxels/s CDF 9/7 3-D fwd float, 1024^3 3-level"
        name = f"CDF 9/7 forward 3-D float, {n}^3, {J} levels, out of place (cdf97_3f_op semantics), one volume per step per GPU"
        def check(b):
            # in-run sanity: a constant volume has no detail -- every coefficient with an odd index on
            # any axis is ~0 and the deepest approximation is the constant times 2^(3J/2)
            c = torch.full((64, 64, 256), 3.0, dtype=torch.float32, device=dev)
            o = torch.empty_like(c)
            dwt.transform3d_op(c, o, 256 * 4, 256 * 64 * 4, 256, 64, 64, 1)
            torch.cuda.synchronize()
            lll = o[0::2, 0::2, 0::2]
            det = float(o[1::2].abs().max())
            return {"constant_volume_ok": bool(det < 1e-5 and float((lll - 3.0 * 2 ** 1.5).abs().max()) < 1e-4)}

    if w == "config4" and args.placements > 1:
        # a resident 2-D batch: through the library's placement-aware allocator, like the headline; attempt 0 =
        # plain first allocations with the library's own search off
        dwt.set_option("place_tries", 1)
        plain = make()
        dwt.tune("cdf97_s", 0, plain[0], plain[1], n * n * 4, nb, n * 4, n, n, J)  # (search off: tile heights only)
        run(plain)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            run(plain)
        torch.cuda.synchronize()
        first_ms = (time.perf_counter() - t0) / 3 * 1e3
        plain = None
        dwt.dwt_util_finish()
        torch.cuda.empty_cache()
        dwt.set_option("place_tries", args.placements)
        p_src, p_dst = dwt.alloc_batch("cdf97_s", nb, n, n, J)
        bufs = (raw_tensor(torch, dev, p_src, (nb, n, n)), raw_tensor(torch, dev, p_dst, (nb, n, n)))
        seeded_images(torch, gen, bufs[0], lo)
        placement = {"by": "dwt_hip_alloc_batch", "attempts": [{"spacer_GiB": 0, "ms_per_step": round(first_ms, 4)}]}
        placement.update(dwt.alloc_batch_report())
    elif w == "config5" and args.placements > 1:
        dwt.set_option("place_tries", 1)
        plain = make()
        run(plain)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            run(plain)
        torch.cuda.synchronize()
        first_ms = (time.perf_counter() - t0) / 3 * 1e3
        plain = None
        dwt.dwt_util_finish()
        torch.cuda.empty_cache()
        dwt.set_option("place_tries", args.placements)
        p_src, p_dst = dwt.alloc_volumes(n, n, n, J)
        bufs = (raw_tensor(torch, dev, p_src, (n, n, n)), raw_tensor(torch, dev, p_dst, (n, n, n)))
        gen.manual_seed(1234 + rank)
        torch.rand((n, n, n), generator=gen, out=bufs[0])
        placement = {"by": "dwt_hip_alloc_volumes", "attempts": [{"spacer_GiB": 0, "ms_per_step": round(first_ms, 4)}]}
        placement.update(dwt.alloc_batch_report())
    else:
        bufs, _spacer, placement = choose_placement(torch, dev, min(args.placements, 3), make, run)

    def step():
        run(bufs)

    def barrier():
        torch.cuda.synchronize()
        plane.barrier()
        torch.cuda.synchronize()
    # measurement is explicit (round 5): tile heights of the 2-D workloads' calls, before the warm-up, untimed
    if w == "config3":
        dwt.tune("cdf53_i", 0, bufs[0], bufs[1], n * n * 4, nb, n * 4, n, n, J)
        dwt.tune("cdf53_i", 1, bufs[1], bufs[2], n * n * 4, nb, n * 4, n, n, J)
    elif w == "config4":
        dwt.tune("cdf97_s", 0, bufs[0], bufs[1], n * n * 4, nb, n * 4, n, n, J)
    if dwt.alloc_batch_note() and "by" in placement and placement["by"].startswith("dwt_hip_alloc"):
        placement["note"] = dwt.alloc_batch_note()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    my_elapsed = elapsed
    elapsed = plane.max(elapsed)
    per_rank = plane.gather_json({"rank": rank, "elapsed_ms_per_step": round(my_elapsed / args.steps * 1e3, 4),
                                  "placement_by": placement.get("by"), "placement_note": placement.get("note")})
    try:
        checks = check(bufs)
    except Exception as e:  # noqa: BLE001
        checks = {"check_error": f"{type(e).__name__}: {e}"}
    if rank != 0:
        return None
    ach = alg * args.steps / elapsed / 1e9
    cfg = {"workload": name, "parallelism": f"batch-sharded x{world}"}
    cfg.update(checks)
    cpu = None
    if world == 1 and not args.no_cpu and w in ("config3", "config5"):
        try:
            cpu = cpu_baseline_config3(n, J) if w == "config3" else cpu_baseline_config5(512)
        except Exception as e:  # noqa: BLE001 -- the checker is optional equipment of the bench
            cpu = {"value": None, "unit": unit, "cores": 0, "kind": "port", "sample": f"failed: {type(e).__name__}: {e}"}
    line = {
        "metric": metric, "value": round(units * args.steps / elapsed / 1e9, 3), "unit": unit, "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": dtype, "data": "synthetic",
        "config": cfg,
        "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                     "kernel": "whole step (all levels) on rank 0: algorithmic bytes / step time"},
        "placement": placement,
        # attempt 0 of the placement choice = plain first allocations (3 untimed steps): what a caller that ignores placement gets
        "value_first_placement": round(units / (placement["attempts"][0]["ms_per_step"] * 1e-3) / 1e9, 3),
    }
    if cpu is not None:
        line["cpu_baseline"] = cpu
    if world > 1:
        line["per_rank"] = per_rank
    return line


def run_rank(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; start it as `python bench.py --gpus N` "
                         f"or `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
    start_deadline(rank, max(30.0, args.timeout - 15.0))

    import torch
    import torch.distributed as dist

    # --force-dist: the distributed branch with whatever WORLD_SIZE is -- on one GPU the rehearsal of everything the first
    # real N > 1 run executes for the first time (RCCL loads, communicator, first collective, gather_json, the batch
    # split as a self send / recv): tests/test_hip_multi.py
    use_dist = world > 1 or args.force_dist
    if use_dist and "MASTER_PORT" not in os.environ:
        os.environ.update({"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port())})
    ndev = torch.cuda.device_count()  # does not initialise the GPU
    if ndev < 1 or not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libdwt_amd has no CPU fallback)")
    # Fewer devices than ranks (a one-GPU box rehearsing the N-rank path): ranks share
    # devices and coordinate over gloo -- RCCL refuses two ranks on one device.  The line is
    # then flagged `devices_shared` and is not a scaling measurement.
    shared = ndev < world
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    os.environ["DWT_HIP_DEVICE"] = str(dev_index)
    dev = torch.device("cuda", dev_index)
    control = "single process"
    plane = Plane(torch, None, dev, False)
    if use_dist:
        # barrier, max-reduce and the batch split only; RCCL bound to this rank's GPU
        plane, control = open_group(torch, dist, dev, dev_index, shared, args.pg_timeout)

    def teardown():
        if use_dist:
            plane.barrier()
            dist.destroy_process_group()

    import libdwt_amd as dwt
    from libdwt_amd.batch import shard_range

    dwt.dwt_util_init()
    for kv in args.opt:
        k, v = kv.split("=")
        dwt.set_option(k, int(v))
    stream = torch.cuda.current_stream()
    dwt.set_stream(stream.cuda_stream)

    if args.workload != "headline":
        out = other_workload(args, dwt, torch, plane, world, rank, dev)
        if out is not None and use_dist:
            out["control_plane"] = control
        finish(rank, out, None, 0, teardown, dev_index)
        return

    n, J, total = args.size, args.levels, args.images
    lo, hi = shard_range(total, rank, world)
    nb = hi - lo
    if nb < 1:
        raise SystemExit(f"bench.py: --images {total} leaves rank {rank} of {world} without an image")
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    img_bytes = n * n * 4
    chunk = args.chunk if args.chunk > 0 else nb

    def run_once(src, dst):
        if args.inplace:
            for k in range(nb):
                dwt.dwt_cdf97_2f_s(dst[k], n * 4, 4, n, n, n, n, J)
        else:
            for k in range(0, nb, chunk):
                c = min(chunk, nb - k)
                dwt.transform2d_batch("cdf97_s", 0, src[k:k + c], dst[k:k + c], img_bytes, c, n * 4, n, n, J)

    # ---- first placement: plain allocations, the library's placement search off (reported beside `value`) ----
    def level0_rate(fn, reps):
        dwt.prof_enable(True)
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        p_ms, p_n = dwt.prof_read()
        dwt.prof_enable(False)
        per = 1 if args.inplace else min(chunk, nb)
        return 2 * 4 * n * n * per / (p_ms / max(p_n, 1) * 1e-3) / 1e9 if p_n else 0.0

    first = None
    placed = args.placements > 1 and not args.inplace
    def tune_shard(src, dst):
        """dwt_hip_tune for the calls run_once makes (measurement is explicit since round 5; untimed, before the warm-up)"""
        if args.inplace:
            return
        t0 = time.perf_counter()
        for c in sorted({min(chunk, nb), nb % chunk or chunk}):
            dwt.tune("cdf97_s", 0, src[:c], dst[:c], img_bytes, c, n * 4, n, n, J)
        return round(time.perf_counter() - t0, 3)

    host_leg = None
    if world == 1 and not args.no_single and not args.inplace:
        try:
            host_leg = host_pointer_leg(torch, dwt, n, J)
        except Exception as e:  # noqa: BLE001
            host_leg = {"error": f"{type(e).__name__}: {e}"}
    dwt.set_option("place_tries", 1)
    src = seeded_images(torch, gen, torch.empty((nb, n, n), device=dev, dtype=torch.float32), lo)
    dst = src.clone() if args.inplace else torch.empty_like(src)
    tune_first_s = tune_shard(src, dst)  # (search off: tile heights only)
    for _ in range(2):
        run_once(src, dst)
    torch.cuda.synchronize()
    if placed:
        t0 = time.perf_counter()
        for _ in range(5):
            run_once(src, dst)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        first = {"gsamples_per_s_this_rank": round(nb * n * n / dt / 1e9, 2), "ms_per_step": round(dt * 1e3, 4),
                 "level0_GBps": round(level0_rate(lambda: run_once(src, dst), 3), 1),
                 "how": "plain first allocations of the shard and of the library's scratch, no placement search (dwt_hip_tune with "
                        "place_tries = 1: tile heights only), 5 steps (untimed for `value`)"}
        # ---- the placed shard: dwt_hip_alloc_batch (destination + scratch chosen by timing), same data ----
        src = dst = None
        dwt.dwt_util_finish()  # drops the plainly placed scratch
        torch.cuda.empty_cache()
        dwt.set_option("place_tries", args.placements)
        t0 = time.perf_counter()
        try:
            p_src, p_dst = dwt.alloc_batch("cdf97_s", nb, n, n, J)
            alloc_s = time.perf_counter() - t0
            src, dst = raw_tensor(torch, dev, p_src, (nb, n, n)), raw_tensor(torch, dev, p_dst, (nb, n, n))
            seeded_images(torch, gen, src, lo)
            placement = {"by": "dwt_hip_alloc_batch", "seconds_total": round(alloc_s, 2)}
            placement.update(dwt.alloc_batch_report())
            note = dwt.alloc_batch_note()
            if note:  # the allocator fell back to plain allocations and says why
                placement["by"] = "plain allocations (dwt_hip_alloc_batch did not search)"
                placement["note"] = note
            placement["how"] = ("untimed: an arena of most of the free memory; the destination tried at every 4 GiB step (one level against "
                                "the source), the LL scratch at every step for the three best destinations (the shard's transform itself); "
                                "the best arrangement kept, the rest of the arena returned; DESIGN s5")
        except Exception as e:  # noqa: BLE001 -- the run goes on with plain allocations and says so
            src = seeded_images(torch, gen, torch.empty((nb, n, n), device=dev, dtype=torch.float32), lo)
            dst = torch.empty_like(src)
            placement = {"by": f"plain allocations (dwt_hip_alloc_batch failed: {type(e).__name__}: {e})"}
        placement["tune_seconds"] = tune_shard(src, dst)
        placement["tune"] = "dwt_hip_tune on the resident shard before the warm-up (untimed): tile heights; the scratch came placed with the batch"
    else:
        placement = {"by": "none (--placements 1 or --inplace): plain first allocations", "tune_seconds": tune_first_s}

    def step():
        run_once(src, dst)

    def barrier():
        torch.cuda.synchronize()
        plane.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    dwt.prof_enable(True)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        step()
        marks[i + 1].record()  # same stream as the launches: per-step HIP-event times
    barrier()
    elapsed = time.perf_counter() - t0
    k_ms, k_launches = dwt.prof_read()
    dwt.prof_enable(False)
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]

    my_elapsed = elapsed
    elapsed = plane.max(elapsed)
    # every rank's own figures into the line (a straggler, or a rank whose allocator fell back, must be visible)
    per_rank = plane.gather_json({
        "rank": rank, "device": dev_index, "images": nb, "elapsed_ms_per_step": round(my_elapsed / args.steps * 1e3, 4),
        "step_ms_min": round(min(step_ms), 4), "step_ms_median": round(statistics.median(step_ms), 4),
        "level0_GBps": round(2 * 4 * n * n * (1 if args.inplace else min(chunk, nb)) / (k_ms / max(k_launches, 1) * 1e-3) / 1e9, 1) if k_launches else None,
        "placement_by": placement.get("by"), "placement_note": placement.get("note"),
        "kept_arrangement_ms": placement.get("kept_arrangement_ms"),
        "first_placement_gsamples_per_s": first["gsamples_per_s_this_rank"] if first else None,
    })

    samples = total * n * n * args.steps
    value = samples / elapsed / 1e9
    alg = algorithmic_bytes(n, n, J)
    # dominant kernel: the level-0 sweep; per launch it reads and writes the whole
    # level-0 region of every image of the call once
    images_per_launch = 1 if args.inplace else min(chunk, nb)
    l0_bytes = 2 * 4 * n * n * images_per_launch
    l0_ms = k_ms / max(k_launches, 1)
    achieved = l0_bytes / (l0_ms * 1e-3) / 1e9 if k_launches else None

    out = None
    if rank == 0:
        traffic, traffic_src = _profile_traffic(l0_bytes, n)
        out = {
            "metric": "Gsamples/s (= % HBM3E BW) CDF 9/7 2-D fwd float, 8192^2 5-level",
            "value": round(value, 3),
            "unit": "Gsamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"CDF 9/7 forward 2-D float, {n}x{n}, {J} levels, fixed batch of {total} device-resident "
                                   f"images sharded b*N//B over {world} GPU(s) ({nb} per GPU), "
                                   + ("in-place entry dwt_cdf97_2f_s per image" if args.inplace
                                      else f"out-of-place entry (dwt_cdf97_2f_s2 semantics), {images_per_launch} images per launch"),
                       "entry": "dwt_cdf97_2f_s" if args.inplace else "dwt_cdf97_2f_s2",
                       "images_total": total, "images_per_gpu": nb, "images_per_launch": images_per_launch,
                       "parallelism": f"batch-sharded x{world}"},
            "hbm_frac_algorithmic": round(value * 1e9 * alg / (n * n) / (HBM_PEAK_GBS * 1e9) / world, 4),
            "step_ms_rank0": {"min": round(min(step_ms), 4), "median": round(statistics.median(step_ms), 4), "n": len(step_ms),
                              "how": "HIP events on the launch stream, rank 0's shard"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1) if achieved else None, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None,
                         "traffic": None, "traffic_from_profile": traffic, "traffic_source": traffic_src,
                         "kernel": "k_fwd_sweep<Cdf97S> level 0",
                         "bytes_per_launch": l0_bytes, "avg_launch_ms": round(l0_ms, 5), "launches": k_launches},
        }
        out["placement"] = placement
        if first is not None:
            # whole-job figure of the first placement: the slowest rank's shard rate x ranks (every rank measured its own)
            rates = [r.get("first_placement_gsamples_per_s") for r in per_rank if r.get("first_placement_gsamples_per_s")]
            out["value_first_placement"] = round((min(rates) if rates else first["gsamples_per_s_this_rank"]) * world, 3)
            out["first_placement"] = first
        if use_dist:
            out["per_rank"] = per_rank
            out["control_plane"] = control
        if shared:
            out["devices_shared"] = True
        STATE["line"] = dict(out)  # from here on the watchdog has something to print
        if world == 1 and not args.inplace and not args.no_sweep and chunk >= nb:
            try:
                out["shard_sweep"], out["projected_scaling"] = shard_sweep(torch, dwt, src, dst, n, J, nb)
            except Exception as e:  # noqa: BLE001
                out["shard_sweep"] = {"error": f"{type(e).__name__}: {e}"}
            STATE["line"] = dict(out)
        if not args.no_single and not args.inplace:
            try:
                out["single_image"] = single_image_stats(torch, dwt, src, dst, n, J)
            except Exception as e:  # noqa: BLE001
                out["single_image"] = {"error": f"{type(e).__name__}: {e}"}
            STATE["line"] = dict(out)
        if world == 1 and not args.no_single and not args.inplace:
            try:
                dwt.transform2d_batch("cdf97_s", 0, src, dst, img_bytes, nb, n * 4, n, n, J)  # coefficients of the whole batch in dst again
                out.update(extra_legs(torch, dwt, src, dst, n, J, nb))
                if host_leg is not None:
                    out["host_pointer"] = host_leg
            except Exception as e:  # noqa: BLE001
                out["extra_legs"] = {"error": f"{type(e).__name__}: {e}"}
            STATE["line"] = dict(out)
        if world == 1 and not args.no_cpu:
            try:
                out["cpu_baseline"] = cpu_baseline(n, J)
            except Exception as e:  # the checker is optional equipment of the bench
                out["cpu_baseline"] = {"value": None, "unit": "Gsamples/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}

    # the side measurement comes LAST and under its own deadline: the scatter / gather of the whole
    # batch over RCCL (never inside `value`); its first execution on real ranks may be the driver's
    split_fn = None
    if use_dist and not args.no_split and not args.inplace:
        if shared:
            split_fn = lambda: {"skipped": "ranks share a device (gloo rehearsal); the RCCL batch split needs one GPU per rank"}
        elif not plane.on_device:
            split_fn = lambda: {"skipped": f"control plane is {control}"}
        else:
            split_fn = lambda: batch_split_times(torch, plane, src, total, n, rank, world, dev)
    finish(rank, out, split_fn, args.split_timeout, teardown, dev_index)


def launcher_selftest(args):
    """CPU rehearsal of the N-rank plumbing (tests/test_bench_launcher.py): the ranks form a gloo
    group and go through the SAME barrier / max-over-ranks / guarded side measurement / one-line /
    teardown code as the GPU run; the side measurement is a real scatter + gather of a small batch
    (libdwt_amd.batch over gloo).  No GPU.  BENCH_FAULT injects the failures the launcher and the
    guards exist for: "exit:R" (rank R dies before the rendezvous), "hang_split:R" (rank R never
    returns from the side measurement), "hang_transform:R" (rank R hangs before there is a line)."""
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    fault, _, frank = os.environ.get("BENCH_FAULT", "").partition(":")
    hit = fault and int(frank or -1) == rank
    if hit and fault == "exit":
        sys.stderr.write(f"rank {rank}: injected failure before the rendezvous\n")
        sys.exit(3)
    start_deadline(rank, max(5.0, args.timeout - 5.0))
    import datetime

    import torch
    import torch.distributed as dist

    from libdwt_amd import batch as B

    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=args.pg_timeout))
    plane = Plane(torch, dist, torch.device("cpu"), False)
    plane.barrier()
    if hit and fault == "hang_transform":
        time.sleep(3600)
    t = plane.max(1.0 + rank)
    lo, hi = B.shard_range(64, rank, world)
    per_rank = plane.gather_json({"rank": rank, "images": hi - lo})
    out = {"selftest": True, "n_gpus": world, "max_over_ranks": t, "images_rank0": hi - lo, "per_rank": per_rank} if rank == 0 else None

    def split():
        if hit and fault == "hang_split":
            time.sleep(3600)
        full = torch.arange(8 * 4 * 4, dtype=torch.float32).reshape(8, 4, 4) if rank == 0 else None
        local = B.scatter_images(full, 8, (4, 4), torch.float32, torch.device("cpu"))
        back = B.gather_images(local * 2, 8)
        plane.barrier()
        return {"round_trip_ok": bool(torch.equal(back, full * 2)) if rank == 0 else None}

    def teardown():
        plane.barrier()
        dist.destroy_process_group()

    finish(rank, out, split, args.split_timeout, teardown)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=8192)
    ap.add_argument("--levels", type=int, default=5)
    ap.add_argument("--images", type=int, default=64, help="images of the whole batch (sharded over the GPUs)")
    ap.add_argument("--chunk", type=int, default=0, help="images per batched call (0 = the rank's whole shard)")
    ap.add_argument("--inplace", action="store_true", help="time the in-place entry dwt_cdf97_2f_s instead of _s2")
    ap.add_argument("--placements", type=int, default=4,
                    help="candidates of (destination, LL scratch) the library's placement-aware allocator may time (1 = plain first allocations)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-single", action="store_true", help="skip the single-image entry timings")
    ap.add_argument("--no-sweep", action="store_true", help="skip the shard-size sweep (N = 1)")
    ap.add_argument("--no-split", action="store_true", help="skip the RCCL scatter/gather timing (N > 1)")
    ap.add_argument("--timeout", type=float, default=600.0,
                    help="whole-run deadline in seconds: the launcher terminates its ranks, a rank prints what it has and leaves")
    ap.add_argument("--split-timeout", type=float, default=90.0, help="deadline of the RCCL batch-split side measurement")
    ap.add_argument("--pg-timeout", type=float, default=300.0,
                    help="process-group rendezvous / collective timeout (the first `import torch` of eight ranks on a fresh box takes minutes)")
    ap.add_argument("--opt", action="append", default=[], help="backend option name=value (cpt, tile_pairs, waves, ...)")
    ap.add_argument("--workload", default="headline", choices=["headline", "config3", "config4", "config5"],
                    help="headline = BASELINE.json's metric (default); config3/4/5 = the other BASELINE configs, same JSON contract")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the torch.distributed branch (gloo + RCCL group, batch split) even with one rank: the one-GPU rehearsal of the N > 1 run")
    ap.add_argument("--selftest-launcher", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around us: start the rank processes (this parent never touches the GPU)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:], timeout_s=args.timeout))
    if args.selftest_launcher:
        if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
            raise SystemExit("bench.py: --gpus does not match WORLD_SIZE")
        return launcher_selftest(args)
    run_rank(args)


if __name__ == "__main__":
    main()
