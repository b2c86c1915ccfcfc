#!/usr/bin/env python
"""Does RCCL admit two ranks on ONE device?  (`bench.py --gpus 2 --share-device` over nccl, or the error text.)
Run under torchrun with 2 ranks on a 1-GPU box; both ranks use cuda:0.  Prints the outcome of init + one all-reduce."""
import os
import sys
import traceback

import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    x = torch.ones(4, device="cuda:0") * (rank + 1)
    dist.all_reduce(x)
    torch.cuda.synchronize()
    print("rank %d: RCCL accepted %d ranks on one device; all-reduce -> %s" % (rank, world, x.tolist()), flush=True)
    dist.destroy_process_group()
except Exception as exc:                                         # noqa: BLE001
    print("rank %d: RCCL refused %d ranks on one device: %s" % (rank, world, "".join(traceback.format_exception_only(type(exc), exc)).strip()[:1500]),
          flush=True)
    sys.exit(0)
