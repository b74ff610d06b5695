"""CPU: `python bench.py --gpus N` must really run N ranks.  The launcher is exercised
with GOLEFT_BENCH_STUB=1 -- a rendezvous over gloo and a trivial step, no engine -- so that what is tested is the
part that decides how many processes take part; and without a GPU the real path must refuse instead of printing
an n_gpus = 1 line."""
import json
import os
import subprocess
import sys

import pytest

from tests import helpers as H

BENCH = os.path.join(H.ROOT, "bench.py")


def _run(args, env_extra, timeout=300):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("n", [2, 3])
def test_launcher_starts_n_ranks(n):
    p = _run(["--gpus", str(n), "--steps", "4", "--warmup", "1"], {"GOLEFT_BENCH_STUB": "1"})
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                               # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["ranks_seen"] == n and d["distinct_devices"] == n
    assert len(d["devices_seen"]) == n and d["launcher"] == "self"
    assert d["value"] is None and d["data"] == "stub"              # never mistaken for a measurement


def test_external_launcher_env_is_respected_and_checked():
    # what torch.distributed.run gives a rank; --gpus that disagrees with WORLD_SIZE is refused
    p = _run(["--gpus", "2", "--steps", "1"], {"GOLEFT_BENCH_STUB": "1", "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in p.stderr
    p = _run(["--gpus", "1", "--steps", "1"], {"GOLEFT_BENCH_STUB": "1"})
    assert p.returncode == 0 and json.loads(p.stdout.strip().splitlines()[-1])["ranks_seen"] == 1


def test_a_failing_rank_fails_the_launch():
    p = _run(["--gpus", "2", "--steps", "-1"], {"GOLEFT_BENCH_STUB": "1", "GOLEFT_BENCH_STUB_FAIL_RANK": "1"})
    assert p.returncode != 0 and "rank 1 exited" in p.stderr


def test_refuses_more_ranks_than_devices():
    """No GPU in this container: --gpus 2 must exit non-zero with a clear message, not print n_gpus: 1."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible")
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {})
    assert p.returncode != 0
    assert "refusing to run 2 ranks" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_the_json_line_is_the_last_thing_on_a_piped_stdout():
    """RCCL prints a version banner through C stdio when it is loaded; on a stdout that is a pipe that text sits in the C
    library's buffer until the process exits -- BEHIND the line rank 0 printed from Python.  bench.flush_c_stdio() empties
    the buffer first (round 5: the two-rank dry runs ended with `Librccl path : ...` as their last line)."""
    import subprocess
    import sys
    code = ("import ctypes, json, sys; sys.path.insert(0, %r)\n"
            "import bench\n"
            "libc = ctypes.CDLL(None); libc.printf(b'ROCm version : banner\\nLibrccl path : banner\\n')\n"
            "bench.flush_c_stdio()\n"
            "print(json.dumps({'metric': 'x'}), flush=True)\n" % H.ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-500:]
    lines = r.stdout.strip().splitlines()
    assert lines[-1] == '{"metric": "x"}' and "Librccl path : banner" in lines[:-1], lines
