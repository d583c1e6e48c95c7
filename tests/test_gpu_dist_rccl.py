"""The native multi-GPU host on RCCL itself with R > 1 ranks (tests/rccl_rank.py: one process per rank, all on cuda:0).

RCCL refuses two ranks of one host on one device, so each rank process is given a host id of its own (NCCL_HOSTID) and the
ranks meet over RCCL's socket transport on the loop-back interface: ncclCommInitRank with R ranks, the grouped ncclSend /
ncclRecv all-to-all of the key bins, ncclAllGather of the packets, and the fixed-slot view round of vh_dist_raycast are the
calls a node with R GPUs would make.  (The loop-back transport of tests/test_gpu_dist_loopback.py covers the same host code
with hipMemcpyAsync in their place and can afford C4 / C5 at size; this file is about the RCCL calls.)  Every rank checks
its shard against its bucket slice of ONE oracle table and its raycast against the oracle's."""
import os
import subprocess
import sys

import pytest

from test_sharding_cpu import _free_port

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def rank_env(rank):
    """Environment of rank `rank`: a host of its own as far as RCCL can tell, sockets on the loop-back interface only."""
    env = dict(os.environ, NCCL_HOSTID=f"voxelhash-test-host-{rank}", NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1",
               NCCL_DEBUG=os.environ.get("NCCL_DEBUG", "ERROR"), GLOO_SOCKET_IFNAME="lo", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def run_ranks(world, batch, sensor, *extra, timeout=600):
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "rccl_rank.py"), str(r), str(world), str(port), str(batch), str(int(sensor)), *extra],
                              env=rank_env(r), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            out, err = p.communicate(timeout=timeout)
            outs.append((p.returncode, out, err))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (rc, out, err) in enumerate(outs):
        assert rc == 0, f"rank {r} exit {rc}\n{out[-2000:]}\n{err[-6000:]}"
        assert f"rank {r}: ok" in out
    return outs


@pytest.mark.parametrize("world,batch,sensor", [(2, 2, True), (2, 1, False), (4, 2, True)])
def test_rccl_ranks_equal_one_oracle_table(torch_cuda, world, batch, sensor):
    run_ranks(world, batch, sensor)


def test_rccl_raycast_round_that_agrees_on_its_capacity(torch_cuda):
    """vh_dist_raycast_auto: the lost counts travel by ncclAllGather, every rank repeats the round with the same capacity."""
    run_ranks(2, 2, True, "raycast_auto")


def test_rccl_c4_at_size(torch_cuda):
    """C4 (BASELINE.json configs): four 640x480 cameras, 2^20 buckets over four ranks, sensor-depth packets, batches of 2."""
    run_ranks(4, 2, True, "size=640x480", "buckets=20", "blocks=15", "exchanges=3", timeout=900)


def test_rccl_eight_ranks(torch_cuda):
    """Eight ranks (C5's split) at a small image size: seven peers per grouped send / receive, an 8-way all-gather."""
    run_ranks(8, 1, True, "buckets=16", "blocks=12", "exchanges=4", timeout=900)
