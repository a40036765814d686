"""Diagnostic: throughput of consecutive forwards issued round-robin on N streams, eager and with each replica's forward replayed as a
captured hipGraph.  usage: python tools/diag/streams_graphs_sweep.py [batch] [max streams]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 4
nets = [bench.build_network(64, dev)[0] for _ in range(NS)]
img, depth = bench.device_inputs(B, 0, dev, "uniform")
streams = [torch.cuda.Stream() for _ in range(NS)]


def run(n_streams, steps):
    with torch.no_grad():
        for i in range(steps):
            k = i % n_streams
            with torch.cuda.stream(streams[k]):
                nets[k](img, nets[k].grid_coord_proj_batch, nets[k].coord_volumes, depth_map_batch=depth)


steps = 60 if B <= 2 else 30
for graphs in (False, True, False, True):
    for n in nets:
        n.enable_graphs(graphs)
    for ns in range(1, NS + 1):
        run(ns, 2 * ns + 4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(ns, steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"B={B} graphs={graphs} {ns} stream(s): {steps * B / dt:8.1f} frames/s  ({dt / steps * 1e3:.3f} ms per forward)", flush=True)
