"""bench.py --gpus N without a launcher (CPU part): the parent process must start the ranks as a child torch.distributed.run
and relay the child's exit code.  Without a GPU every rank stops with bench.py's "needs an MI355X" message -- which is exactly
what proves the relay here: the message comes from the CHILD ranks and the parent's exit code is non-zero."""
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_self_launch_spawns_ranks_and_relays_exit_code():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("CPU-only check (the GPU twin is tests/test_gpu_bench_contract.py::test_bench_self_launch)")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode != 0
    assert "needs an MI355X" in out.stderr                      # raised inside the ranks, i.e. the launcher ran them
    assert "launch with torch.distributed.run" not in out.stderr


def test_bench_parent_makes_no_gpu_call_before_launching():
    """Static check of the ordering: in main(), self_launch() is reached before msnets_amd / the library are imported."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    assert main.index("self_launch(args.gpus)") < main.index("import msnets_amd")
    body = src[src.index("def self_launch"):src.index("def main():")]
    assert "os.exec" not in body and "subprocess.Popen" in body
