"""Test-only stand-ins for the sharded path's backends: the same interface as
voxelhashing_demo_amd.dist.HipShard / HipViewTable on the CPU oracle (buffers are CPU tensors).
They let the per-rank logic of dist.py (sharded_step, sharded_raycast, the transports) run on
CPU with gloo, world_size 2, and serve as the reference the HIP shards are compared with."""
import numpy as np

from voxelhashing_demo_amd.dist import VIEW_RECORD_BYTES, ShardPlan, _view_params


class OracleShard:
    """The same interface on the CPU oracle (tests only; the buffers are CPU tensors)."""

    def __init__(self, oracle_module, params, width, height, semantics, plan: ShardPlan, rank: int, capacity: int,
                 batch: int = 1):
        import torch
        self.plan, self.rank, self.capacity, self.batch = plan, rank, capacity, batch
        self.table = oracle_module.OracleTable(params, width, height, semantics, bucket_range=plan.bucket_range(rank))
        self.packet_floats = P = 32 + width * height
        R, B = plan.world, batch
        self.bins_send = torch.zeros((R, B, capacity, 4), dtype=torch.int32)
        self.bins_recv = torch.zeros((R, B, capacity, 4), dtype=torch.int32)
        self.packet = torch.zeros((B, P), dtype=torch.float32)
        self.packets = torch.zeros((R, B, P), dtype=torch.float32)

    def generate(self, b: int, pose, verts):
        import torch
        self.table.set_pose(pose)
        bins, packet = self.table.generate_keys(np.asarray(verts), self.rank, self.plan.world, self.capacity)
        self.bins_send[:, b] = torch.from_numpy(bins)
        self.packet[b] = torch.from_numpy(packet)

    def apply(self, b: int):
        self.table.reset_mutexes()
        self.table.insert_bins(self.bins_recv[:, b].contiguous().numpy())
        self.table.integrate_packets(self.packets[:, b].contiguous().numpy())

    def generate_all(self, poses, verts_list, depth_list=None):     # the oracle always ships float planes
        for b in range(self.batch):
            self.generate(b, poses[b], verts_list[b])

    def apply_all(self):
        for b in range(self.batch):
            self.apply(b)

    def export_views(self, poses, capacity: int, t_min: float = 0.1, t_max: float = 5.0):
        import torch
        recs, counts = [], []
        for pose in poses:
            r, n = self.table.export_view(pose, capacity, t_min, t_max)
            recs.append(r)
            counts.append(n)
        packed = np.concatenate(recs) if recs else np.zeros((0, VIEW_RECORD_BYTES), np.uint8)
        return torch.from_numpy(packed), torch.tensor(counts, dtype=torch.int32)


class OracleViewTable:
    """The same on the CPU oracle (tests only)."""

    def __init__(self, oracle_module, params, width, height, semantics, world: int = 1, capacity: int = 0):
        import torch
        self.table = oracle_module.OracleTable(_view_params(params), width, height, semantics)
        self.recv = torch.zeros((max(1, world * capacity), VIEW_RECORD_BYTES), dtype=torch.uint8)

    def render(self, count: int, pose, t_min: float = 0.1, t_max: float = 5.0):
        import torch
        dropped = self.table.import_view(self.recv[:count].numpy())
        assert dropped == 0
        return torch.from_numpy(self.table.raycast(pose, t_min, t_max))
