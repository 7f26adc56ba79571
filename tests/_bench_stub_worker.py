"""Stand-in for bench.py under torch.distributed.run (CPU only): checks the launcher's environment and argument forwarding;
rank 0 prints one JSON line, every rank some noise that must not reach the launcher's stdout."""
import json
import os
import sys

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
print("noise from rank %d" % rank)
if "--fail" in sys.argv:
    sys.exit(3)
if rank == 0:
    print(json.dumps({"metric": "stub", "n_gpus": world, "argv": sys.argv[1:], "master": os.environ.get("MASTER_ADDR"),
                      "ipc": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}))
