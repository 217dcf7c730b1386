"""The double-buffered reduce of bench.py over RCCL itself, as far as one GPU allows: a process group of one rank,
asynchronous reduces issued from two streams of different priority, every frame retired in order.

python tools/nccl_pipeline_check.py [PORT]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
from rt_amd import sharding
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = sys.argv[1] if len(sys.argv) > 1 else "29533"
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
streams = [torch.cuda.Stream(priority=0), torch.cuda.Stream(priority=-1)]
bufs = [torch.zeros(1 << 20, device="cuda") for _ in range(2)]
done = []
pipe = sharding.ReducePipeline(dist, 2, bufs, on_done=lambda k, b: done.append((k, float(b[0].item()))), rank=0,
                               context=lambda k: torch.cuda.stream(streams[k % 2]))  # world forced to 2: collectives are issued
for k in range(6):
    pipe.acquire(k)
    with torch.cuda.stream(streams[k % 2]):
        bufs[k % 2].fill_(k + 1.0)
        pipe.submit(k)
pipe.drain()
dist.barrier()
torch.cuda.synchronize()
print(done)
assert done == [(k, k + 1.0) for k in range(6)]
dist.destroy_process_group()
print("nccl single-rank pipeline ok")
