"""CPU: `bench.py --gpus N` starts its own ranks.  Without a launcher (no WORLD_SIZE) the parent -- before it
touches torch / the GPU -- runs `python -m torch.distributed.run --nproc-per-node N bench.py <same args>` as a child and exits
with its status; under a launcher, WORLD_SIZE must equal --gpus.  (On this GPU-less container the ranks themselves stop at
"needs a HIP device": that message, once per rank, is the evidence that N ranks were started.  The GPU-side rehearsal of the
same entry point is tests/test_gpu_shards.py::test_bench_two_rank_control_flow_rehearsal.)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = dict(os.environ, **kw)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        if k not in kw:
            env.pop(k, None)
    return env


def test_world_size_must_equal_gpus():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=_env(WORLD_SIZE="3", RANK="0", LOCAL_RANK="0"),
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 2
    assert "WORLD_SIZE=3 but --gpus 2" in out.stderr
    assert "Traceback" not in out.stderr


def test_gpus_n_starts_n_ranks_as_a_child_and_relays_the_status():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present: covered by the gloo rehearsal in test_gpu_shards.py")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], env=_env(),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode != 0                                    # the child's status, relayed
    assert out.stderr.count("bench.py needs a HIP device") >= 2    # one per rank: two ranks ran
    assert "torch.distributed" in out.stderr or "ChildFailedError" in out.stderr
