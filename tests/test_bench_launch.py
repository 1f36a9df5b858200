"""`python bench.py --gpus N` must be a valid command on its own: with WORLD_SIZE unset the bench starts
its N ranks as a child torchrun BEFORE touching the GPU, forwards rank 0's JSON line and the children's exit code.
No GPU here: LMI_BENCH_LAUNCH_CHECK=1 makes the ranks do the rendezvous + one gloo all-reduce only.  The whole
bench through the same path, on the card, is `tests/test_gpu_sharded.py::test_bench_self_launch_two_ranks_gloo`."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(gpus, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update({"LMI_BENCH_LAUNCH_CHECK": "1", "OMP_NUM_THREADS": "1"}, **(extra_env or {}))
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "1", "--warmup", "0",
                           "--n", "1000", "--nq", "10", "--nb", "2"],   # --n / --nb: prefixes of torchrun's own options
                          env=env, capture_output=True, text=True, timeout=300)


def test_bench_gpus2_self_launches_and_forwards_json():
    r = _run(2)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    assert json.loads(lines[0]) == {"launch_check": True, "world": 2, "sum": 3, "rccl_ranks_seen": 2, "lib_comm": None}
    assert "2 ranks up over gloo" in r.stderr and "rank 1: pid" in r.stderr   # the roll call, before anything is built


def test_bench_child_failure_is_the_exit_code():
    r = _run(2, {"LMI_BENCH_LAUNCH_CHECK_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_watchdog_ends_a_rank_that_never_reaches_the_first_collective():
    """the first real multi-rank RCCL run will be the driver's -- a stuck bring-up must end within the watchdog's
    limit with a non-zero code and a message, not hang until the lease's."""
    import time

    t0 = time.time()
    r = _run(2, {"LMI_BENCH_LAUNCH_CHECK_HANG_RANK": "1", "LMI_BENCH_WATCHDOG_S": "5"})
    assert r.returncode != 0
    assert time.time() - t0 < 120
    assert "WATCHDOG rank" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_library_communicator_branch_runs_once_on_the_cpu():
    r = _run(2, {"LMI_BENCH_LIBCOMM": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert line["lib_comm"] == ["stub-comm", 0, 2] and line["rccl_ranks_seen"] == 2
