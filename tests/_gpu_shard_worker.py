"""Worker for the GPU test of the sharded fit: R ranks share cuda:0 (rehearsal transport over gloo/host staging)
and must reproduce the single-GPU GaussianProcess and the oracle."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "scikit-gpuppy_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import skgpuppy_amd as sk  # noqa: E402
from skgpuppy_amd.distributed import HostStagedComm, ShardedGaussianProcess, TorchComm  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    N, d, M = int(sys.argv[1]), int(sys.argv[2]), 333
    light = len(sys.argv) > 3 and sys.argv[3] == "light"     # large N: against the single-GPU path only (the oracle's LU inverse of
                                                             # that size on two host threads would take minutes)
    rng = np.random.RandomState(20240 + N + d)
    x = rng.uniform(0, 10, (N, d))
    t = np.sin(0.3 * x.sum(1)) + 0.1 * rng.randn(N)
    xs = rng.uniform(0, 10, (M, d))
    theta = np.log(np.array([2.0, 0.01] + [0.04] * d))
    torch.cuda.set_device(0)
    # transport: "host" (default) = broadcasts staged through the host with every stream synchronised around them; "gloo-device" = the product's
    # TorchComm on a gloo group with DEVICE tensors -- gloo orders itself against the posting stream by events, like RCCL, so the schedule's own
    # event edges (buffer reuse, head before tail, copies into L) are what keeps the ranks correct
    transport = sys.argv[4] if len(sys.argv) > 4 else "host"
    comm = HostStagedComm() if transport == "host" else TorchComm()
    gp = ShardedGaussianProcess(x, t, theta, device=torch.device("cuda", 0), comm=comm)
    mean, var = gp.estimate_many(xs)
    ok = True
    if rank == 0:
        print("panel message: %s, transport: %s" % ("head + tail" if gp.layout.split else "whole", transport))
    if rank == 0:
        ref = sk.GaussianProcess(x, t, sk.GaussianCovariance(), theta.copy())
        m1, v1 = ref.estimate_many(xs)
        e1 = max(np.abs(mean - m1).max(), np.abs(var - v1).max())
        if light:
            print("sharded vs single-GPU: %.3e   (%d ranks, %d panels)" % (e1, dist.get_world_size(), gp.layout.npanels))
            ok = e1 < 1e-9
        else:
            og = orc.OracleGP(x, t, theta)
            om, ov = og.estimate_many(xs)
            e2 = max(np.abs(mean - om).max(), np.abs(var - ov).max())
            print("sharded vs single-GPU: %.3e   sharded vs oracle: %.3e" % (e1, e2))
            ok = e1 < 1e-10 and np.allclose(mean, om, rtol=1e-6, atol=1e-9) and np.allclose(var, ov, rtol=1e-6, atol=2e-9)
    # propagation on the replicated factor: calls sharded over the ranks, every rank gets every result
    us = np.array([[5.0] * d, [4.0] * d, [6.5] * d, x[3]])
    Ss = [0.01 * np.eye(d)] * 4
    pm, pv = gp.propagate_many(us, Ss)
    if rank == 0:
        up = sk.UncertaintyPropagationApprox(ref)
        og_ = None if light else orc.OracleGP(x, t, theta)
        for i in range(4):
            want = up.propagate_GA(us[i], Ss[i])
            ok = ok and abs(pm[i] - want[0]) < 1e-9 and abs(pv[i] - want[1]) < 1e-9
            if not light:
                owant = orc.approx_propagate(og_, us[i], Ss[i])
                ok = ok and abs(pm[i] - owant[0]) < 1e-8 and abs(pv[i] - owant[1]) < 2e-8
        print("sharded propagate_many vs single-GPU and oracle:", ok)
    # ONE propagation shared by the ranks: row panels of K^-1, 4 + 2 d partial sums, one all-reduce (last row)
    for i in (0, 3):
        for via in ("solve", "kinv"):      # right-hand sides dealt to the ranks (no K^-1 anywhere) / row panels of K^-1
            sm, sv = gp.propagate_GA_sharded(us[i], Ss[i], via=via)
            if rank == 0:
                ok = ok and abs(sm - pm[i]) < 1e-9 and abs(sv - pv[i]) < 1e-9
                if not light:
                    owant = orc.approx_propagate(og_, us[i], Ss[i])
                    ok = ok and abs(sm - owant[0]) < 1e-8 and abs(sv - owant[1]) < 2e-8
        em, ev = gp.propagate_exact_sharded(us[i], Ss[i])
        if rank == 0:
            ewant = sk.UncertaintyPropagationExact(ref).propagate_GA(us[i], Ss[i])
            ok = ok and abs(em - ewant[0]) < 1e-9 and abs(ev - ewant[1]) < 1e-9
            if not light:
                eo = orc.exact_propagate(og_, us[i], Ss[i])
                ok = ok and abs(em - eo[0]) < 1e-8 and abs(ev - eo[1]) < 2e-8
    if rank == 0:
        print("row-sharded propagate_GA / Exact vs call-sharded, single-GPU and oracle:", ok)
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    gp.close()
    dist.destroy_process_group()
    sys.exit(0 if flag.item() == 1.0 else 1)


if __name__ == "__main__":
    main()
