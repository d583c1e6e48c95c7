import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np, torch
import voxelhashing_demo_amd as V
from voxelhashing_demo_amd import dist as vdist, synth
W,H=640,480
plan=vdist.ShardPlan(1<<20,1)
stream=torch.cuda.Stream()
with torch.cuda.stream(stream):
    sh=vdist.HipShard(V.default_params(numBuckets=1<<20,numVoxelBlocks=1<<18),W,H,1,plan,0,38400,batch=1,stream=stream)
    poses=synth.camera_loop(500)[:20]; prims=synth.room_primitives()
    verts=[synth.render_room_verts(p,W,H,prims,device='cuda') for p in poses]
    torch.cuda.synchronize()
    def timeit(name, fn, n=200):
        for i in range(10): fn(i)
        torch.cuda.synchronize(); t=time.perf_counter()
        for i in range(n): fn(i)
        t1=time.perf_counter()-t
        torch.cuda.synchronize(); t2=time.perf_counter()-t
        print(f"{name}: host {1e6*t1/n:.1f} us, total {1e6*t2/n:.1f} us")
    t=sh.table
    timeit("set_pose", lambda i: t.set_pose(poses[i%20]))
    timeit("generate", lambda i: sh.generate(0,poses[i%20],verts[i%20]))
    sh.bins_recv.copy_(sh.bins_send); sh.packets[0].copy_(sh.packet)
    timeit("reset", lambda i: t.reset_mutexes())
    timeit("insert_bins", lambda i: (t.reset_mutexes(), t.insert_bins(sh._recv_b[0],1,38400,bin_stride=38400)))
    timeit("integrate_packets", lambda i: t.integrate_packets(1, sh._packets_b[0], packet_stride=sh.packet_floats))
    timeit("apply", lambda i: sh.apply(0))
    timeit("vh_integrate", lambda i: t.integrate(poses[i%20],verts[i%20]))
