#!/usr/bin/env python3
"""Which levels' skip blocks should V2VProgram.run fork onto its side stream?  Whole forward, one stream of
issue, eager and hipGraph replay, for batch 1 and 8 and every suffix set of levels (0 = 64^3 ... 4 = 4^3).
    python tools/diag/fork_sweep.py [--batches 1,8]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import build_network, device_inputs  # noqa: E402


def timed(net, img, depth, steps):
    with torch.no_grad():
        for _ in range(3):
            net(img, net.grid_coord_proj_batch, net.coord_volumes, depth_map_batch=depth)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = net(img, net.grid_coord_proj_batch, net.coord_volumes, depth_map_batch=depth)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, out[0].clone()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="1,8")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    net, _ = build_network(64, dev)
    sets = ["", "4", "3,4", "2,3,4", "1,2,3,4", "0,1,2,3,4", "0", "0,1"]
    for B in [int(b) for b in args.batches.split(",")]:
        img, depth = device_inputs(B, 0, dev, "uniform")
        ref = None
        for fs in sets:
            os.environ["SCENEEGO_FORK_LEVELS"] = fs
            net.enable_graphs(False)
            e_ms, kp = timed(net, img, depth, 30 if B == 1 else 10)
            net._graphs.clear()
            net.enable_graphs(True)
            g_ms, kpg = timed(net, img, depth, 30 if B == 1 else 10)
            net.enable_graphs(False)
            net._graphs.clear()
            if ref is None:
                ref = kp
            print(f"B={B} fork levels [{fs:9s}]  eager {e_ms:7.3f} ms ({B / e_ms * 1e3:7.1f} frames/s)   hipGraph {g_ms:7.3f} ms "
                  f"({B / g_ms * 1e3:7.1f} frames/s)   max|joints - unforked| {float((kp - ref).abs().max()):.1e} / graph {float((kpg - ref).abs().max()):.1e}",
                  flush=True)
    os.environ.pop("SCENEEGO_FORK_LEVELS", None)


if __name__ == "__main__":
    main()
