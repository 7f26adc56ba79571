"""CPU, world_size 2 (gloo): the multi-GPU panel schedule of skgpuppy_amd.distributed on CPU tensors."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.parametrize("n,pb,world,mode", [(700, 1, 2, "bcast"), (1000, 2, 2, "split"), (390, 1, 3, "split"), (900, 1, 2, "split"),
                                             (1000, 2, 2, "split-whole"), (390, 1, 3, "bcast-whole")])
def test_panel_cholesky_gloo(n, pb, world, mode):
    """mode: transport ("bcast" one broadcast per message part, "split" scatter + all-gather) and, with "-whole", one message per panel
    instead of head (the next panel's square rows) + tail"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29600 + (n + 31 * len(mode)) % 97),
           os.path.join(ROOT, "tests", "_gloo_worker.py"), str(n), str(pb), mode]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "max |L - Lref|" in r.stdout


def test_stalled_status_reaches_every_rank_without_a_hang():
    """only the owner of a stalled panel sees GPX_INFO_STALLED in its status word.  GpxOps.finish() must not raise on that
    rank alone (the others would block in the all-reduce until the backend's timeout): the raw word goes through panel_cholesky's
    max_int and every rank gets it; ShardedGaussianProcess.refit raises on all of them, before the jitter retry."""
    import inspect
    from skgpuppy_amd import distributed as D
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3",
           "--master-addr", "127.0.0.1", "--master-port", "29711", os.path.join(ROOT, "tests", "_gloo_worker.py"), "390", "1", "bcast-stall"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("stalled status agreed") == 3
    assert "raise RuntimeError" not in inspect.getsource(D.GpxOps.finish)                      # the product's finish() hands the word on
    src = inspect.getsource(D.ShardedGaussianProcess.refit)
    assert src.index("INFO_STALLED") < src.index("if info > 0")                   # raised collectively, not answered with jitter


def test_layout_ownership():
    from skgpuppy_amd.distributed import PanelLayout
    lay = PanelLayout(65536, 8)
    assert lay.npad == 65536 and lay.nblk == 512 and lay.npanels == 64
    assert sorted(sum((lay.owned(r) for r in range(8)), [])) == list(range(64))
    assert lay.blocks(63) == (504, 512)
    lay = PanelLayout(1000, 3, panel_blocks=2)
    assert lay.npad == 1024 and lay.npanels == 4 and lay.blocks(3) == (6, 8)
    assert [lay.owner(p) for p in range(4)] == [0, 1, 2, 0]


def test_row_shards_and_partial_combination():
    """Host side of the row-sharded Approx propagation (last row): the shards tile [0, n) in 128-aligned
    panels, and partial sums over the shards -- computed here from the oracle's K^-1 -- combine to the oracle's
    propagate_GA (the device side of the same partials is covered by the -m gpu two-rank test)."""
    import numpy as np
    from oracle import oracle as orc
    from skgpuppy_amd.distributed import combine_approx_partials, rhs_shards, row_shards
    for nvec, world in ((9, 2), (17, 8), (3, 8), (65, 3)):
        sh = rhs_shards(nvec, world)
        assert len(sh) == world and sh[0][0] == 0 and sh[-1][1] == nvec
        assert all(a[1] == b[0] for a, b in zip(sh, sh[1:])) and max(b - a for a, b in sh) - min(b - a for a, b in sh) <= 1
    for n, world in ((1000, 3), (128, 2), (16384, 8), (5, 4)):
        sh = row_shards(n, world)
        assert sh[0][0] == 0 and sh[-1][1] == n and all(a[1] == b[0] for a, b in zip(sh, sh[1:]))
        assert all((lo % 128 == 0 or lo == n) and (hi % 128 == 0 or hi == n) for lo, hi in sh)
    for n, world in ((1000, 3), (16384, 8), (65536, 8), (130, 4)):
        sh = row_shards(n, world, triangular=True)
        assert sh[0][0] == 0 and sh[-1][1] == n and all(a[1] == b[0] for a, b in zip(sh, sh[1:]))
        assert all((lo % 128 == 0 or lo == n) and (hi % 128 == 0 or hi == n) and hi >= lo for lo, hi in sh)
        if n >= 16384:      # equal area of the lower triangle within a few per cent
            area = [hi * hi - lo * lo for lo, hi in sh]
            assert max(area) < 1.15 * min(area)
    rng = np.random.RandomState(5)
    n, d = 300, 3
    x = rng.uniform(0, 10, (n, d))
    t = np.sin(0.3 * x.sum(1)) + 0.1 * rng.randn(n)
    theta = np.log(np.array([2.0, 0.01] + [0.04] * d))
    og = orc.OracleGP(x, t, theta)
    u = np.array([5.0, 4.0, 6.0])
    S = np.diag([0.01, 0.02, 0.005])
    C, J, H = orc.cjh(og, u)
    J = J.reshape(n, d)
    tr = np.einsum("iab,ba->i", H.reshape(n, d, d), S)
    beta = og.beta()
    tot = np.zeros(4 + 2 * d)
    for lo, hi in row_shards(n, 3):
        r = slice(lo, hi)
        KC = og.Kinv[r].dot(C)
        part = [beta[r].dot(C[r]), beta[r].dot(tr[r]), C[r].dot(KC), KC.dot(tr[r])]
        for k in range(d):
            part += [J[r, k].dot(og.Kinv[r].dot(J[:, k])), beta[r].dot(J[r, k])]
        tot += np.array(part)
    mu, var, _s2, _rest = combine_approx_partials(tot, S, np.exp(theta[0]), np.exp(theta[1]))
    om, ov = orc.approx_propagate(og, u, S)
    assert mu + og.meant == pytest.approx(om, abs=1e-10) and var == pytest.approx(ov, abs=1e-9)
