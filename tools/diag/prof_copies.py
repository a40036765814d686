"""Which aten op issues the device-to-device copies seen as __amd_rocclr_copyBuffer in the kernel trace (diagnostic)."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

net, sd = bench.build_network(64, "cuda:0")
img, depth = bench.device_inputs(8, 0, "cuda:0", "uniform")
with torch.no_grad():
    for _ in range(3):
        net(img, net.grid_coord_proj_batch, net.coord_volumes, depth_map_batch=depth)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        net(img, net.grid_coord_proj_batch, net.coord_volumes, depth_map_batch=depth)
        torch.cuda.synchronize()
evs = prof.events()
copies = [e for e in evs if "copy" in e.name.lower() or "memcpy" in e.name.lower() or "Memcpy" in e.name]
print("copy-like events:", len(copies))
from collections import Counter
cnt = Counter()
for e in copies:
    p = e.cpu_parent
    chain = []
    while p is not None and len(chain) < 4:
        chain.append(p.name)
        p = p.cpu_parent
    cnt[(e.name[:40], " <- ".join(chain), str(getattr(e, "input_shapes", ""))[:60])] += 1
for k, v in cnt.most_common(30):
    print(v, k)
