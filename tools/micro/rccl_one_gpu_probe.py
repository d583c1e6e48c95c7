#!/usr/bin/env python3
"""Can RCCL run two ranks on ONE GPU when each rank claims to be another host (NCCL_HOSTID) and the ranks talk over the loop-back
socket?  (RCCL refuses two ranks of one host on one device: "Duplicate GPU detected".)  Probe only."""
import os
import subprocess
import sys

if len(sys.argv) > 1:
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, init_method=f"tcp://127.0.0.1:{port}")
    x = torch.full((1024,), float(rank + 1), device="cuda")
    dist.all_reduce(x)
    a = torch.arange(world * 4, dtype=torch.int32, device="cuda") + 100 * rank
    b = torch.empty_like(a)
    dist.all_to_all_single(b, a)
    torch.cuda.synchronize()
    print(f"rank {rank}: all_reduce -> {x[0].item()}, all_to_all -> {b.tolist()}", flush=True)
    dist.destroy_process_group()
    sys.exit(0)

world = 2
import socket
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
procs = []
for r in range(world):
    env = dict(os.environ, NCCL_HOSTID=f"probe-host-{r}", NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1", NCCL_DEBUG="WARN",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs.append(subprocess.Popen([sys.executable, __file__, str(r), str(world), str(port)], env=env))
rc = 0
for p in procs:
    try:
        rc |= p.wait(timeout=240)
    except subprocess.TimeoutExpired:
        p.kill()
        rc |= 99
print("probe exit", rc)
sys.exit(rc)
