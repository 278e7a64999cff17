"""One rank of a sharded forward (run as WORLD_SIZE child processes by tests/test_parallel_gpu.py, or by hand):
with backend "gloo" every rank sits on cuda:0 (one-GPU rehearsal of the multi-GPU data path), with "nccl" (= RCCL) rank r
sits on cuda:LOCAL_RANK and the exchange is the device-side all_gather_into_tensor; each rank runs its shard_range slice of a
seeded net_deploy batch through forward -> decode_outputs -> sharded_post_process and saves the gathered detections of the
WHOLE batch.   python scripts/sharded_forward_check.py <out.npz> <total> <size> <conf> [gloo|nccl]"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccnet_amd import synth
from ccnet_amd.network.net_deploy import Net
from ccnet_amd.parallel import shard_range, sharded_post_process
from ccnet_amd.utils import bbox

out_path, total, size, conf = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
backend = sys.argv[5] if len(sys.argv) > 5 else "gloo"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = "cuda:%d" % (int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0)
torch.cuda.set_device(dev)
if backend == "nccl":
    dist.init_process_group("nccl", device_id=torch.device(dev))
else:
    dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
net = Net(0.33, 0.5, 6, "silu")
net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=0, bn_calib="net_deploy"))
net = net.to(dev).eval().set_precision("bf16")
lo, hi = shard_range(total, rank, world)
x = synth.synth_images(total, size, size, seed=77)[lo:hi].contiguous().to(dev)
info = [[i, size + 3 * i, size - 2 * i] for i in range(total)]
for _ in range(2):  # the second call runs on the carried-over row estimate
    dec = bbox.decode_outputs(net(x), [size, size])
    dets = sharded_post_process(dec, 6, [size, size], info[lo:hi], conf, 0.6, total_images=total)
assert len(dets) == total
np.savez(out_path, counts=np.array([d.shape[0] for d in dets]), rows=np.concatenate(dets, 0) if dets else np.zeros((0, 7), np.float32))
dist.barrier()
dist.destroy_process_group()
