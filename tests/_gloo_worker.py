"""Worker for tests/test_distributed_gloo.py: runs the product's panel schedule (skgpuppy_amd.distributed.
panel_cholesky, PanelLayout, TorchComm) on CPU tensors over the gloo backend with a torch-CPU `ops`, and checks the
factor every rank ends up with against numpy's Cholesky of the oracle's Gram matrix."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "scikit-gpuppy_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
from skgpuppy_amd.distributed import INFO_STALLED, TILE, PanelLayout, TorchComm, panel_cholesky  # noqa: E402
from oracle import oracle as orc  # noqa: E402


class CpuOps(object):
    """Reference implementation of the ops interface on CPU tensors (test infrastructure)."""

    def __init__(self, x, theta, layout, rank):
        self.layout, self.x, self.theta, self.rank = layout, x, theta, rank
        npad = layout.npad
        self.stage = [torch.empty(layout.message_elems(0), dtype=torch.float64) for _ in range(3)]     # message slots
        self.L = torch.full((npad, npad), float("nan"), dtype=torch.float64)   # NaN: unbuilt regions must never be read
        self.calls = []

    def _K(self, rows, cols):
        n = self.layout.n
        K = torch.zeros((len(rows), len(cols)), dtype=torch.float64)
        r_in, c_in = rows < n, cols < n
        with np.errstate(divide="ignore"):
            full = orc.gram_ij(self.x[rows[r_in]], self.x[cols[c_in]], self.theta)
            vt = np.exp(self.theta[1])
        K[np.ix_(r_in, c_in)] = torch.as_tensor(full)
        eq = torch.as_tensor(rows[:, None] == cols[None, :])
        K[eq & torch.as_tensor(r_in[:, None] & c_in[None, :])] += vt
        K[eq & ~torch.as_tensor(r_in[:, None] & c_in[None, :])] = 1.0           # identity padding
        return K

    def build_panel(self, p):
        b0, b1 = self.layout.blocks(p)
        rows = np.arange(b0 * TILE, self.layout.npad)
        cols = np.arange(b0 * TILE, b1 * TILE)
        self.L[b0 * TILE:, b0 * TILE:b1 * TILE] = self._K(rows, cols)
        self.calls.append(("build", p))

    def factor_panel(self, p, prev):
        if prev is not None:
            self._update(p, prev)
        b0, b1 = self.layout.blocks(p)
        c0, c1 = b0 * TILE, b1 * TILE
        D = torch.linalg.cholesky(torch.tril(self.L[c0:c1, c0:c1]) + torch.tril(self.L[c0:c1, c0:c1], -1).T)
        self.L[c0:c1, c0:c1] = D
        if c1 < self.layout.npad:
            self.L[c1:, c0:c1] = torch.linalg.solve_triangular(D, self.L[c1:, c0:c1].T, upper=False).T
        self.calls.append(("factor", p))
        # the message: rows below the square | the square (the product packs inverted blocks and the diagonal behind it)
        buf = self._slot(p)
        buf.fill_(float("nan"))
        w = c1 - c0
        nlow = (self.layout.npad - c1) * w
        buf[:nlow].copy_(self.L[c1:, c0:c1].reshape(-1))
        buf[nlow:nlow + w * w].copy_(D.reshape(-1))
        buf[nlow + w * w:].zero_()

    def _update(self, q, p):
        pb0, pb1 = self.layout.blocks(p)
        qb0, qb1 = self.layout.blocks(q)
        r0 = qb0 * TILE
        A = self.L[r0:, pb0 * TILE:pb1 * TILE]
        B = self.L[r0:qb1 * TILE, pb0 * TILE:pb1 * TILE]
        assert not torch.isnan(A).any(), "update read a panel that was never received"
        self.L[r0:, r0:qb1 * TILE] -= A @ B.T
        self.calls.append(("update", q, p))

    def update_panels(self, qs, p):
        for q in qs:
            self._update(q, p)

    def _slot(self, p):
        return self.stage[p % len(self.stage)][:self.layout.message_elems(p)]

    def message(self, p, part):
        lo, hi = self.layout.part_range(p, part)
        assert hi > lo and part in self.layout.parts(p)
        self.calls.append(("message", p, part))
        return self._slot(p)[lo:hi]

    def adopt(self, p, part, buf, work):
        work.wait()
        self.calls.append(("adopt", p, part))
        if self.layout.owner(p) == self.rank:
            return
        b0, b1 = self.layout.blocks(p)
        c0, c1 = b0 * TILE, b1 * TILE
        w, hr = c1 - c0, self.layout.head_rows(p)
        if part == "head":
            assert buf.numel() == hr * w
            self.L[c1:c1 + hr, c0:c1] = buf.view(hr, w)
        else:
            below = self.layout.npad - c1
            rest = buf[:(below - hr) * w].view(below - hr, w)
            self.L[c1 + hr:, c0:c1] = rest
            self.L[c0:c1, c0:c1] = buf[(below - hr) * w:(below - hr) * w + w * w].view(w, w)

    stall_on_rank = None      # test hook: this rank's status word reports a timed-out hand-off (GPX_INFO_STALLED)

    def finish(self):
        return INFO_STALLED if self.rank == self.stall_on_rank else 0


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n, d, pb = int(sys.argv[1]), 3, int(sys.argv[2])
    rng = np.random.RandomState(11)
    x = rng.uniform(0, 5, (n, d))
    theta = np.log(np.array([1.5, 0.05, 0.3, 0.2, 0.4]))
    mode = sys.argv[3] if len(sys.argv) > 3 else "bcast"
    stall = mode.endswith("-stall")
    layout = PanelLayout(n, world, panel_blocks=pb, split=not mode.endswith("-whole"))
    ops = CpuOps(x, theta, layout, rank)
    if stall:
        ops.stall_on_rank = world - 1
    # split_bytes=1: every message whose length the world size divides travels as scatter + all-gather
    info = panel_cholesky(ops, layout, rank, TorchComm(split_bytes=1 if mode.startswith("split") else 1 << 40))
    if stall:
        # ONE rank's status word says "stalled": no rank throws before the collective, every rank learns it from the all-reduce
        # (a rank that raised alone would leave the others blocked in it) -- the caller then raises on all of them
        assert info == INFO_STALLED, info
        print("rank %d: stalled status agreed" % rank)
        dist.destroy_process_group()
        sys.exit(0)
    assert info == 0
    with np.errstate(divide="ignore"):
        K = orc.gram(x, theta)
    Lref = np.linalg.cholesky(K)
    L = torch.tril(ops.L[:n, :n]).numpy()
    err = np.abs(L - Lref).max()
    # ownership / schedule properties
    built = [c[1] for c in ops.calls if c[0] == "build"]
    factored = [c[1] for c in ops.calls if c[0] == "factor"]
    assert built == layout.owned(rank) and factored == layout.owned(rank), (built, factored)
    for c in ops.calls:
        if c[0] == "update":
            assert layout.owner(c[1]) == rank and c[2] < c[1]
    # every panel's parts are posted and adopted by every rank in the same order: head (the next square's rows) before tail
    want = [(p, part) for p in range(layout.npanels) for part in layout.parts(p)]
    assert [c[1:] for c in ops.calls if c[0] == "message"] == want and [c[1:] for c in ops.calls if c[0] == "adopt"] == want
    if layout.split and layout.npanels > 1:
        assert ("head" in layout.parts(0)) and layout.parts(layout.npanels - 1) == ("tail",)
    nupd = sum(1 for c in ops.calls if c[0] == "update")
    assert nupd == sum(q for q in layout.owned(rank)), (nupd, layout.owned(rank))   # every earlier panel exactly once
    ok = torch.tensor([1.0 if err < 1e-10 else 0.0])
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if rank == 0:
        print("max |L - Lref| = %.3e on %d ranks, %d panels" % (err, world, layout.npanels))
    dist.destroy_process_group()
    sys.exit(0 if ok.item() == 1.0 else 1)


if __name__ == "__main__":
    main()
