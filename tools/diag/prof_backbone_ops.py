"""Per-convolution device time of the folded backbone, by input / weight shape (diagnostic; torch.profiler)."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sceneego_amd import pose_resnet  # noqa: E402

net = pose_resnet.get_pose_net(None).to("cuda:0").eval()
fb = pose_resnet.FoldedBackbone(net)
x = torch.randn(8, 3, 256, 256, device="cuda:0")
with torch.no_grad():
    for _ in range(5):
        y = fb(x)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        y = fb(x)
        torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key in ("aten::miopen_convolution", "aten::miopen_convolution_transpose", "aten::bmm", "aten::index_select", "aten::mm"):
        shp = e.input_shapes[:2]
        rows.append((e.device_time_total, e.count, e.key.replace("aten::", ""), shp))
tot = 0
for t, n, k, shp in sorted(rows, reverse=True):
    ci = shp[0][1] if shp and len(shp[0]) > 1 else 0
    print(f"{t:9.1f} us total  {n:2d} x {t / n:7.1f}  {k:28s} {shp}")
    tot += t
print(f"{tot:9.1f} us in these ops")
