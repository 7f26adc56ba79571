"""Worker of test_panel_transport_on_rccl_multi_gpu (launched by torch.distributed.run, one rank per GPU): the panel transport
TorchComm.broadcast over RCCL with scatter + all-gather forced (split_bytes=1), two staging slots reused round-robin as in the
factorisation, every received buffer compared with what the source sent.  A stale chunk (the race the in-place all-gather
removes) shows up as the PREVIOUS message's values in a slot."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scikit-gpuppy_amd"))
from skgpuppy_amd.distributed import TorchComm  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)))
    dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
    for split_bytes in (1, 1 << 60):
        comm = TorchComm(split_bytes=split_bytes)
        n = world * 4096 * 33
        slots = [torch.zeros(n, dtype=torch.float64, device="cuda") for _ in range(2)]
        side = torch.cuda.Stream()
        works = []
        for p in range(12):
            src = p % world
            buf = slots[p % 2]
            with torch.cuda.stream(side):
                if rank == src:
                    buf.copy_(torch.arange(n, dtype=torch.float64, device="cuda") * 1e-3 + (p + 1))
                w = comm.broadcast(buf, src)
            w.wait()
            got = buf.clone()
            torch.cuda.synchronize()
            want = torch.arange(n, dtype=torch.float64, device="cuda") * 1e-3 + (p + 1)
            assert torch.equal(got, want), "rank %d panel %d (split_bytes=%d): %d wrong values" % (
                rank, p, split_bytes, int((got != want).sum()))
            works.append(w)
    assert comm.max_int(rank) == world - 1
    dist.barrier()
    if rank == 0:
        print("panel transport over RCCL ok (%d ranks)" % world)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
