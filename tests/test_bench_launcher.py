"""bench.py's N-rank plumbing on CPU: `python bench.py --gpus N` with no launcher around it
must start N rank processes itself (the parent makes no GPU call), relay rank 0's JSON line
with n_gpus == N, and fail when --gpus disagrees with WORLD_SIZE.  The GPU workload itself is
replaced by the hidden --selftest-launcher leg (gloo group, barrier, max-over-ranks)."""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    return env


def test_gpus_flag_starts_the_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--selftest-launcher"], env=_env(), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2
    assert line["max_over_ranks"] == 2.0  # rank 1 reported 1.0 + 1
    assert line["images_rank0"] == 32      # 64 images, b*N//B


def test_gpus_flag_must_match_world_size():
    env = _env()
    env.update({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--selftest-launcher"], env=env, capture_output=True, text=True,
                       timeout=120)
    assert r.returncode != 0
    assert "WORLD_SIZE" in (r.stderr + r.stdout)


def test_parent_does_not_import_torch():
    """The launching parent must not initialise anything GPU-side: it never imports torch."""
    src = open(BENCH).read()
    head = src[:src.index("def run_rank")]
    launch = head[head.index("def launch_ranks"):head.index("def cpu_baseline")]
    assert "import torch" not in launch
    # module level: no torch import either
    assert "\nimport torch" not in src.split("def algorithmic_bytes")[0]
