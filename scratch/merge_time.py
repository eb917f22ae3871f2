# cost of the banded-mode hit merge (host tags -> device gather -> device sort) with a 1-rank nccl group
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
torch.cuda.init(); torch.cuda.set_device(0)
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', RANK='0', WORLD_SIZE='1')
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
from kevlar_amd import bandmerge, _lib
_lib.load(); _lib.require_device()
n = 1_150_000
rng = np.random.default_rng(1)
r = np.sort(rng.integers(0, 7_500_000, size=n)).astype(np.uint32); o = rng.integers(0, 70, size=n).astype(np.uint32)
a = rng.integers(0, 30, size=(n, 3)).astype(np.uint8)
for _ in range(4):
    t0 = time.perf_counter()
    out = bandmerge.allgather_hits_device(r, o, a, torch.device('cuda', 0))
    torch.cuda.synchronize()
    print('merge of %d hits: %.2f ms' % (n, (time.perf_counter() - t0) * 1e3), len(out[0]))
dist.destroy_process_group()
