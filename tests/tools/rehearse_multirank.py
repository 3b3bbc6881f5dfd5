"""Multi-rank rehearsal of the sharded flow on the real engine (several ranks sharing the GPUs
that are there, gloo backend): work-weighted strips + pipelined gather + steps alternating over
two streams, every frame of every step compared with the oracle.
    python -m torch.distributed.run --nproc-per-node 3 --master-addr 127.0.0.1 tests/tools/rehearse_multirank.py
"""
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import torch.distributed as dist

from stepth_amd import MODE_RING_FIRST_HIT, Engine, synth
from stepth_amd.sharding import EngineStripWorker, GatherPipeline, StripBalancer, sharded_depth

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
dev = torch.device("cuda", torch.cuda.current_device())
dist.init_process_group("gloo")
W, H, F, STEPS = 320, 182, 6, 9
pairs = [synth.stereo_pair(W, H, 48, seed=900 + i)[:2] for i in range(F)]
frames = [(torch.from_numpy(np.ascontiguousarray(l)).to(dev), torch.from_numpy(np.ascontiguousarray(r)).to(dev)) for l, r in pairs]
torch.cuda.synchronize()
engines = [Engine(dev.index) for _ in range(2)]
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
for e, s in zip(engines, streams):
    with torch.cuda.stream(s):
        e.use_torch_stream()
workers = [EngineStripWorker(e, [36] * 3, max_steps=48, tile=(1, 1), mode=MODE_RING_FIRST_HIT) for e in engines]
# expected depth maps: the oracle (the CPU restatement of the reference), not the engine
from oracle import oracle as O
want = [O.depth_from_additional(l, r, [36] * 3, 48, (1, 1))["depth"] for l, r in pairs]
bal = StripBalancer(H, world, align=2, min_rows=8, lag=2)
pipes = [GatherPipeline(), GatherPipeline()]
history, bad = [], 0
for step in range(STEPS):
    slot = step % 2
    history.append(list(bal.bounds))
    with torch.cuda.stream(streams[slot]):
        depth, mx = sharded_depth(workers[slot], frames, H, W, dst=0, balancer=bal, pipeline=pipes[slot])
    if rank == 0 and depth is not None:  # the maps of step - 2 (same slot)
        torch.cuda.synchronize()
        got = depth.cpu().numpy()
        bad += sum(not np.array_equal(got[f], want[f]) for f in range(F))
for k in range(2):
    slot = (STEPS + k) % 2
    with torch.cuda.stream(streams[slot]):
        depth = pipes[slot].flush()
    if rank == 0:
        torch.cuda.synchronize()
        got = depth.cpu().numpy()
        bad += sum(not np.array_equal(got[f], want[f]) for f in range(F))
gathered = [None] * world
dist.all_gather_object(gathered, history)
same = all(h == gathered[0] for h in gathered)
if rank == 0:
    print(f"ranks {world}: mismatching frames {bad}, cut points identical on all ranks: {same}, "
          f"cuts moved: {history[0] != history[-1]}; final heights {[b - a for a, b in history[-1]]}")
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if (bad == 0 and same) else 1)
