"""Latency of the direct-write gather (hdn_gather_offsets_oneshot) between two processes sharing GPU 0, eager and as a
hipGraph replay, beside a plain device copy of the same size.  (RCCL refuses two ranks on one device, so there is no RCCL
column on a one-GPU box; the 8-GPU comparison is the driver's.)   python tools/experiments/exp_oneshot_gather.py"""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

WORKER = r"""
import os, sys, time, json
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("gloo", rank=rank, world_size=world)
from hdn_amd import dist as hdist
g = hdist.OneShotGather.from_process_group(64, dev)
x = torch.randn(64, 8, device=dev)
def timed(fn, n):
    torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
for _ in range(20): g.all_gather(x)
eager = timed(lambda: g.all_gather(x), 2000)
torch.cuda.synchronize(); dist.barrier()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    for _ in range(10): y = g.all_gather(x)
rep = timed(graph.replay, 300) / 10
z = torch.empty(world * 64, 8, device=dev)
cp = timed(lambda: z[:64].copy_(x), 2000)
if rank == 0:
    print(json.dumps({"world": world, "bytes_per_rank": 2048, "oneshot_us_eager": eager, "oneshot_us_in_graph": rep, "device_copy_us_eager": cp,
                      "status": g.status()}), flush=True)
dist.barrier(); g.destroy(); dist.destroy_process_group()
"""

s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
procs = [subprocess.Popen([sys.executable, "-c", WORKER % {"root": ROOT}],
                          env=dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0"))
         for r in range(2)]
sys.exit(max(p.wait(timeout=300) for p in procs))
