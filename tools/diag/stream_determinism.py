"""Diagnostic: is the float32 forward bitwise reproducible run to run, and across HIP streams (MIOpen / rocBLAS keep per-stream
handles and may select different kernels)?"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda:0")
net = bench.build_network(64, dev)[0]
img, depth = bench.device_inputs(2, 0, dev, "floor")


def fwd(stream=None):
    with torch.no_grad():
        if stream is None:
            out = net(img, net.grid_coord_proj_batch, net.coord_volumes, depth_map_batch=depth)
        else:
            with torch.cuda.stream(stream):
                out = net(img, net.grid_coord_proj_batch, net.coord_volumes, depth_map_batch=depth)
    torch.cuda.synchronize()
    return out[0].clone(), out[1].clone()


a = fwd(); b = fwd()
print("default stream, run 1 vs run 2: joints", float((a[0] - b[0]).abs().max()), " backbone features", float((a[1] - b[1]).abs().max()))
s1 = torch.cuda.Stream()
c = fwd(s1); d = fwd(s1)
print("side stream,    run 1 vs run 2: joints", float((c[0] - d[0]).abs().max()), " backbone features", float((c[1] - d[1]).abs().max()))
print("default vs side stream        : joints", float((a[0] - c[0]).abs().max()), " backbone features", float((a[1] - c[1]).abs().max()),
      " (features max", float(a[1].abs().max()), ")")
