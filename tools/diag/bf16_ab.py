#!/usr/bin/env python3
"""config 3 (bf16 storage, B=32) forward time and per-shape conv launch times with the library named by SCENEEGO_HIP_LIB.
    SCENEEGO_HIP_LIB=sceneego_amd/libse_x.so python tools/diag/bf16_ab.py [batch]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import build_network, device_inputs  # noqa: E402
from sceneego_amd import _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
net, _ = build_network(64, dev)
net.set_v2v_dtype("bf16")
net.set_backbone_dtype("bf16")
img, depth = device_inputs(B, 0, dev, "uniform")
with torch.no_grad():
    for _ in range(3):
        kp = net(img, net.grid_coord_proj_batch, net.coord_volumes, depth_map_batch=depth)[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        kp = net(img, net.grid_coord_proj_batch, net.coord_volumes, depth_map_batch=depth)[0]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 8
    _lib.start_profile()
    for _ in range(3):
        net(img, net.grid_coord_proj_batch, net.coord_volumes, depth_map_batch=depth)
    torch.cuda.synchronize()
    prof = _lib.stop_profile()
print(f"{os.path.basename(_lib.LIB_PATH)}: B={B} bf16 forward {dt * 1e3:.3f} ms = {B / dt:.1f} frames/s; joints checksum {float(kp.double().sum()):.6f}")
for k in sorted(prof, key=lambda k: -sum(prof[k])):
    v = prof[k]
    if k[0] != "stage" and sum(v) / 3 > 0.05:
        print(f"   {str(k):44s} {len(v) // 3:3d}/step avg {sum(v) / len(v):8.4f} ms  per-step {sum(v) / 3:8.4f} ms")
print("   stages:", {k[1]: round(sum(v) / len(v), 3) for k, v in prof.items() if k[0] == "stage"})
