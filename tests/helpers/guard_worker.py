"""Stand-in for a data-parallel worker under probnmn.launch_guard (tests/test_launch_guard.py): joins a gloo process
group, all-reduces once per "step" and beats; modes (argv[1]):
  ok        every attempt works
  hang0     attempt 0: the last rank stops before the first collective (the others then block in it); attempt 1 works
  hang      every attempt hangs
  crash0    attempt 0: the last rank exits 7 after the first step; attempt 1 works"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "probnmn-clevr_amd"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from probnmn import launch_guard, parallel  # noqa: E402

mode = sys.argv[1]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
launch_guard.beat()
total = 0.0
for step in range(3):
    if rank == world - 1 and (mode == "hang" or (mode == "hang0" and launch_guard.attempt() == 0)):
        time.sleep(1e6)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    total += float(t)
    launch_guard.beat()
    if rank == world - 1 and mode == "crash0" and launch_guard.attempt() == 0:
        os._exit(7)
if rank == 0:
    print("not the json line")
    print(json.dumps({"value": total, "serial": parallel.serial_collectives(), "attempt": launch_guard.attempt()}), flush=True)
dist.destroy_process_group()
