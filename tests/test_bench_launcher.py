"""bench.py's N-rank plumbing on CPU: `python bench.py --gpus N` with no launcher around it
must start N rank processes itself (the parent makes no GPU call), relay rank 0's JSON line
with n_gpus == N, and fail when --gpus disagrees with WORLD_SIZE.  The GPU workload itself is
replaced by the hidden --selftest-launcher leg, which goes through the SAME group / barrier /
max-over-ranks / guarded side measurement / one-line / teardown code as the GPU run (gloo).
BENCH_FAULT injects the failures the launcher's watchdog and the guards exist for."""
import json
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "BENCH_FAULT")}
    env.update(extra)
    return env


def _json_lines(text):
    return [json.loads(ln) for ln in text.splitlines() if ln.startswith("{")]


def test_gpus_flag_starts_the_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--selftest-launcher"], env=_env(), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = _json_lines(r.stdout)
    assert len(lines) == 1  # ONE line
    line = lines[0]
    assert line["n_gpus"] == 2
    assert line["max_over_ranks"] == 2.0  # rank 1 reported 1.0 + 1
    assert line["images_rank0"] == 32      # 64 images, b*N//B
    assert line["batch_split"] == {"round_trip_ok": True}  # scatter + gather over gloo inside the guard
    # every rank's own figures reach the line (host tensors over gloo), in rank order
    assert line["per_rank"] == [{"rank": 0, "images": 32}, {"rank": 1, "images": 32}]


def test_three_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--selftest-launcher"], env=_env(), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    line = _json_lines(r.stdout)[-1]
    assert line["n_gpus"] == 3 and line["max_over_ranks"] == 3.0 and line["batch_split"] == {"round_trip_ok": True}
    assert [r["images"] for r in line["per_rank"]] == [22, 21, 21]  # 64 images, image b -> rank b*3//64


def test_gpus_flag_must_match_world_size():
    env = _env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--selftest-launcher"], env=env, capture_output=True, text=True,
                       timeout=120)
    assert r.returncode != 0
    assert "WORLD_SIZE" in (r.stderr + r.stdout)


def test_rank_dying_before_rendezvous_fails_fast():
    """Rank 1 exits with code 3 before it joins the group: rank 0 would wait in the rendezvous for
    the process-group timeout.  The parent must notice, stop rank 0 and return non-zero quickly."""
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--selftest-launcher", "--timeout", "200", "--pg-timeout", "200"],
                       env=_env(BENCH_FAULT="exit:1"), capture_output=True, text=True, timeout=120)
    took = time.time() - t0
    assert r.returncode != 0
    assert took < 45, took
    assert not _json_lines(r.stdout)
    assert "injected failure" in r.stderr  # the dead rank's log is relayed


def test_hang_in_the_batch_split_keeps_the_transform_line():
    """A rank that never returns from the side measurement (the first-ever RCCL point-to-point on the
    driver's node could): the line with the transform's numbers must still come out, once, exit 0."""
    for who in ("1", "0"):
        r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--selftest-launcher", "--split-timeout", "4", "--timeout", "120"],
                           env=_env(BENCH_FAULT="hang_split:" + who), capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        lines = _json_lines(r.stdout)
        assert len(lines) == 1
        assert lines[0]["max_over_ranks"] == 2.0
        assert "error" in lines[0]["batch_split"] and "4 s" in lines[0]["batch_split"]["error"]


def test_hang_before_any_line_ends_at_the_deadline():
    """A rank wedged inside the timed part: nothing to print; the run deadline ends it, non-zero."""
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--selftest-launcher", "--timeout", "12"],
                       env=_env(BENCH_FAULT="hang_transform:1"), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert time.time() - t0 < 60
    assert not _json_lines(r.stdout)


def test_rank_logs_are_written():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--selftest-launcher"], env=_env(), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0
    for k in (0, 1):
        assert os.path.exists(os.path.join(ROOT, "gpurun_out", f"bench_rank{k}.log"))


def test_parent_does_not_import_torch():
    """The launching parent must not initialise anything GPU-side: it never imports torch."""
    src = open(BENCH).read()
    launch = src[src.index("def launch_ranks"):src.index("STATE = {")]
    assert "import torch" not in launch
    # module level: no torch import either
    assert "\nimport torch" not in src.split("def algorithmic_bytes")[0]
