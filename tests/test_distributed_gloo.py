"""CPU, world_size 2 (gloo): the multi-GPU panel schedule of skgpuppy_amd.distributed on CPU tensors."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.parametrize("n,pb,world", [(700, 1, 2), (1000, 2, 2), (390, 1, 3)])
def test_panel_cholesky_gloo(n, pb, world):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29600 + n % 97),
           os.path.join(ROOT, "tests", "_gloo_worker.py"), str(n), str(pb)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "max |L - Lref|" in r.stdout


def test_layout_ownership():
    from skgpuppy_amd.distributed import PanelLayout
    lay = PanelLayout(65536, 8)
    assert lay.npad == 65536 and lay.nblk == 512 and lay.npanels == 64
    assert sorted(sum((lay.owned(r) for r in range(8)), [])) == list(range(64))
    assert lay.blocks(63) == (504, 512)
    lay = PanelLayout(1000, 3, panel_blocks=2)
    assert lay.npad == 1024 and lay.npanels == 4 and lay.blocks(3) == (6, 8)
    assert [lay.owner(p) for p in range(4)] == [0, 1, 2, 0]
