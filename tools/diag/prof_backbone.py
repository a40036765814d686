"""torch.profiler view of the folded backbone: which aten op issues which device kernel / copy (diagnostic)."""
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
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=60, max_name_column_width=60,
                                                         max_shapes_column_width=70))
