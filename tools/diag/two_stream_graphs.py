"""Diagnostic: two-stream pipelining with each replica's forward replayed as a captured hipGraph."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nets = [bench.build_network(64, dev)[0] for _ in range(2)]
img, depth = bench.device_inputs(B, 0, dev, "uniform")
streams = [torch.cuda.Stream() for _ in range(2)]


def run(n_streams, steps):
    with torch.no_grad():
        for i in range(steps):
            k = i % n_streams
            with torch.cuda.stream(streams[k]):
                nets[k](img, nets[k].grid_coord_proj_batch, nets[k].coord_volumes, depth_map_batch=depth)


for graphs in (False, True, False, True):
    for n in nets:
        n.enable_graphs(graphs)
    for ns in (1, 2):
        run(ns, 6)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(ns, 40)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"graphs={graphs} {ns} stream(s): {40 * B / dt:8.1f} frames/s  ({dt / 40 * 1e3:.3f} ms per forward)")
