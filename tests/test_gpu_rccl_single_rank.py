"""The RCCL ("nccl") branches of the N > 1 path on the hardware this box has: a process group of ONE rank on cuda:0.
What it covers: ``init_process_group("nccl", device_id=...)`` as bench.py calls it, the padded
``all_gather_into_tensor`` of dist.all_gather_rows on DEVICE tensors, the float64 MAX / int64 SUM all-reduces bench.py uses for
its timing and cell count, and the barrier -- the same calls, dtypes and buffers an 8-rank run makes; what it cannot cover is
more than one rank (the 2-rank rendezvous and shard layout run under gloo in tests/test_dist_gloo.py / test_bench_launch.py).
Runs in a child process so that the process group never leaks into the test session."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys
sys.path.insert(0, os.environ["RIBCA_ROOT"])
import torch
import torch.distributed as tdist
from multiplexed_image_annotator_amd import dist
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
tdist.init_process_group("nccl", device_id=dev)
assert tdist.get_backend() == "nccl" and tdist.get_world_size() == 1
n = 1001
g = torch.Generator().manual_seed(3)
local = torch.rand((n, 33), generator=g).to(dev)
full = dist.all_gather_rows(local, n, force_collective=True)
ok_gather = bool(torch.equal(full, local)) and full.data_ptr() != local.data_ptr()
t = torch.tensor([1.25], dtype=torch.float64, device=dev)
tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
c = torch.tensor([99972], dtype=torch.int64, device=dev)
tdist.all_reduce(c, op=tdist.ReduceOp.SUM)
tdist.barrier()
torch.cuda.synchronize()
out = {"backend": tdist.get_backend(), "world": tdist.get_world_size(), "gather_ok": ok_gather, "max": float(t.item()), "sum": int(c.item())}
tdist.destroy_process_group()
print(json.dumps(out))
"""


@pytest.mark.gpu
def test_rccl_single_rank_collectives():
    import json
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", RIBCA_ROOT=ROOT,
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out == {"backend": "nccl", "world": 1, "gather_ok": True, "max": 1.25, "sum": 99972}
