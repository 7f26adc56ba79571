"""CPU-only: `bench.py --gpus N` (N > 1) started as a plain script launches torch.distributed.run as a child process, forwards
rank 0's JSON line and the exit code; the parent never imports torch."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(extra, env_extra=None):
    env = dict(os.environ, GPX_BENCH_LAUNCH_TARGET=os.path.join(ROOT, "tests", "_bench_stub_worker.py"), GPX_BENCH_MASTER_PORT="29533")
    env.pop("WORLD_SIZE", None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True, timeout=300, env=env)


def test_self_launch_forwards_rank0_line_and_arguments():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                       # ONE JSON line, the ranks' other output went to stderr
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["argv"] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]
    assert d["master"] == "127.0.0.1" and d["ipc"] == "0"
    assert "noise from rank" in r.stderr


def test_self_launch_propagates_failure_of_the_child():
    r = _run(["--gpus=2", "--fail"])
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_requested_gpus_parsing_without_heavy_imports():
    code = ("import sys; sys.argv=['bench.py']; sys.path.insert(0, %r)\n"
            "import importlib.util\n"
            "src = open(%r).read().split('import numpy as np')[0]\n"
            "ns = {'__name__': 'bench_head', '__file__': %r}\n"
            "exec(compile(src, 'bench_head', 'exec'), ns)\n"
            "assert ns['requested_gpus'](['--steps', '2']) == 1\n"
            "assert ns['requested_gpus'](['--gpus', '8']) == 8 and ns['requested_gpus'](['--gpus=4']) == 4\n"
            "assert 'torch' not in sys.modules and 'numpy' not in sys.modules\n"
            "print('ok')\n") % (ROOT, os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "bench.py"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
