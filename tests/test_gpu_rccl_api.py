"""The collectives stepth_amd/sharding.py issues, on a one-rank RCCL group (backend "nccl"):
several ranks cannot share one GPU under RCCL, so this checks only that every call form used
by the sharded flow is accepted by the backend -- int32 MAX all_reduce, async gather straight
into views of one receive buffer from a side stream, the pipelined per-destination gathers of
dst="frames", the frame split's gathers, all_gather_into_tensor, barrier.  The
multi-rank logic itself is covered by the gloo tests and tests/tools/rehearse_multirank.py."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

SCRIPT = r'''
import os, torch, torch.distributed as dist
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.arange(10, dtype=torch.int32, device=dev)
dist.all_reduce(x, op=dist.ReduceOp.MAX)
assert x.tolist() == list(range(10))
c = torch.ones(256, dtype=torch.int32, device=dev)
dist.all_reduce(c, op=dist.ReduceOp.SUM)
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
send = torch.full((3, 5, 7), 9, dtype=torch.uint8, device=dev)
recv = torch.empty((1, 3, 5, 7), dtype=torch.uint8, device=dev)
with torch.cuda.stream(torch.cuda.Stream()):
    w = dist.gather(send, list(recv.unbind(0)), dst=0, async_op=True)
    w.wait()
    ok = bool((recv == 9).all())
torch.cuda.synchronize()
assert ok
# dst="frames" with the gathers pipelined (round 3): per destination one async gather of the
# frames it owns (an index-selected, contiguous part), all started before any is waited for
works = []
for d in range(1):
    part = send[[0, 2]].contiguous()
    recv_d = torch.empty((1, 2, 5, 7), dtype=torch.uint8, device=dev)
    works.append((dist.gather(part, list(recv_d.unbind(0)), dst=d, async_op=True), part, recv_d))
for w, _p, r_ in works:
    w.wait()
    assert bool((r_ == 9).all())
# frame split: the finished maps and their int32 maxima gather to one rank
mx = torch.tensor([3, 4, 5], dtype=torch.int32, device=dev)
rm = torch.empty((1, 3), dtype=torch.int32, device=dev)
dist.gather(mx, list(rm.unbind(0)), dst=0)
assert rm[0].tolist() == [3, 4, 5]
r2 = torch.empty((1, 105), dtype=torch.uint8, device=dev)
dist.all_gather_into_tensor(r2.view(-1), send.view(-1))
assert bool((r2 == 9).all())
dist.barrier()
dist.destroy_process_group()
print("RCCL-API-OK")
'''


def test_collective_call_forms_on_a_one_rank_rccl_group():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0 and "RCCL-API-OK" in r.stdout, (r.stdout + r.stderr)[-2000:]
