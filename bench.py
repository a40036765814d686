#!/usr/bin/env python3
"""Headline benchmark: frames/s of the full VoxelNetwork_depth forward (256x256 image + 1024x1280 depth -> 15x3 joints).

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 launched by torch.distributed.run, one
rank per GPU (RCCL).  W untimed warm-up steps, then EXACTLY K steps bracketed by barrier + synchronize on both
sides, MAX over ranks, rank 0 prints ONE JSON line.  A step = one forward over one batch of synthetic input
already resident in HBM (+ the all-gather of the joints when N>1).  Weak scaling: every rank runs
--batch frames (BASELINE.json configs[1]: batch 8, fp32, 64^3).

Extra objects on the line (N=1 only does the CPU leg):
  roofline     — dominant V2V kernel (3x3x3 conv 32->32 at 64^3, 9 launches/step): algorithmic FLOP per launch /
                 average launch duration from HIP events recorded on the launch stream inside the timed region,
                 against the dense f32 MFMA peak (157.3 TFLOP/s); "stage" adds the whole-V2V figures.
  cpu_baseline — the CPU oracle (oracle/sceneego_oracle.py, same ATen CPU ops as the reference) timed on this
                 box's host cores on a bounded sample (B=2 frames, 1 warm-up + 1 timed forward).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32 MFMA peak = vector f32 peak
BF16_MFMA_PEAK_TFLOPS = 2500.0
HBM_PEAK_GBS = 8000.0
# algorithmic work of V2V + soft-argmax per frame at 64^3 fp32
V2V_GFLOP_PER_FRAME = {64: 299.1, 128: 2393.0}
V2V_GB_PER_FRAME = {64: 1.372, 128: 10.98}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="frames per GPU per step (BASELINE configs[1]: 8)")
    ap.add_argument("--volume-size", type=int, default=64)
    ap.add_argument("--depth-kind", default="uniform", choices=["uniform", "floor"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--v2v-dtype", default="fp32", choices=["fp32", "bf16"],
                    help="fp32: BASELINE configs[1] (default, the headline); bf16: configs[2] (bf16 storage, f32 accumulate)")
    ap.add_argument("--backbone-dtype", default="fp32", choices=["fp32", "bf16"], help="MIOpen backbone precision (config 3: bf16)")
    ap.add_argument("--no-extras", action="store_true", help="skip the side measurement of BASELINE configs[2] (bf16, B=32)")
    ap.add_argument("--dump-kernel-events", action="store_true", help="per-shape conv launch times to stderr")
    ap.add_argument("--graphs", action="store_true", help="replay the forward as a captured hipGraph (implies --no-kernel-events)")
    return ap.parse_args()


def build_network(volume_size, device):
    from sceneego_amd import load_config, synth
    from sceneego_amd.voxel_net_depth import VoxelNetwork_depth
    cfg = load_config()
    cfg.model.volume_size = volume_size
    net = VoxelNetwork_depth(cfg, device="cpu", verbose=False)
    sd = synth.make_state_dict(net.state_dict(), seed=0)
    net.load_state_dict(sd, strict=True)
    return net.to(device).eval(), sd


def device_inputs(batch, rank, device, kind):
    g = torch.Generator(device=device)
    g.manual_seed(1234 + rank)
    img = torch.randn((batch, 3, 256, 256), generator=g, device=device, dtype=torch.float32)
    if kind == "uniform":
        depth = torch.rand((batch, 1024, 1280), generator=g, device=device, dtype=torch.float32) * 2.7 + 0.3
    else:
        from sceneego_amd import synth
        depth = synth.make_inputs(1234 + rank, batch, "floor")[1].to(device)
    return img, depth


def effective_cores():
    """Host cores this process may really use: affinity mask, capped by the cgroup CPU quota if one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return n


def pmc_traffic(batch, G):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (None if not measured for
    this shape).  PMC counters cannot be read from inside this process; the passes are tools/pmc_round.sh."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc.json")) as f:
            d = json.load(f)
        if d.get("batch") == batch and d.get("volume_size") == G:
            return d["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    return None


def cpu_baseline(sd, volume_size):
    from oracle import sceneego_oracle as O
    from sceneego_amd import synth
    torch.set_num_threads(effective_cores())
    const = O.Constants(os.path.join(ROOT, "sceneego_amd", "calibration", "fisheye.calibration_05_08.json"), G=volume_size)
    frames = 2
    img, depth = synth.make_inputs(1234, frames, "uniform")
    O.forward(sd, const, img, depth)                         # warm-up at the timed shapes (oneDNN primitive creation)
    times = {}
    t0 = time.perf_counter()
    O.forward(sd, const, img, depth, times=times)
    dt = time.perf_counter() - t0
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            model = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "")
    except OSError:
        pass
    return {"value": round(frames / dt, 4), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle/sceneego_oracle.py forward, B={frames} frames, {volume_size}^3, fp32, 1 warm-up + 1 timed",
            "seconds": round(dt, 3), "cpu": model, "stage_seconds": {k: round(v, 3) for k, v in times.items()}}


def main():
    args = parse()
    from sceneego_amd import _lib, dist as sdist
    rank, world, device = sdist.init_from_env()
    assert world == args.gpus or world == 1, (world, args.gpus)
    _lib.load()
    net, sd = build_network(args.volume_size, device)
    img, depth = device_inputs(args.batch, rank, device, args.depth_kind)
    G = args.volume_size
    bf16 = args.v2v_dtype == "bf16"
    if bf16:
        net.set_v2v_dtype("bf16")
    if args.backbone_dtype == "bf16":
        net.set_backbone_dtype("bf16")
    if args.graphs:
        args.no_kernel_events = True
        net.enable_graphs(True)

    def step():
        kp = net(img, net.grid_coord_proj_batch, net.coord_volumes, depth_map_batch=depth)[0]
        return sdist.all_gather_joints(kp)

    with torch.no_grad():
        for _ in range(args.warmup):
            out = step()
        torch.cuda.synchronize()
        sdist.barrier()
        torch.cuda.synchronize()
        if not args.no_kernel_events:
            _lib.start_profile()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        torch.cuda.synchronize()
        sdist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    prof = _lib.stop_profile() if not args.no_kernel_events else {}
    dt = sdist.max_over_ranks(dt, device)
    assert tuple(out.shape) == (args.batch * world, 15, 3) and bool(torch.isfinite(out).all())

    if rank != 0:
        return
    ms_per_step = dt / args.steps * 1e3
    frames = args.batch * world * args.steps
    line = {
        "metric": "frames/sec VoxelNetDepth forward (256x256 img+depth, 64^3 grid)" if G == 64 else
                  f"frames/sec VoxelNetDepth forward (256x256 img+depth, {G}^3 grid)",
        "value": round(frames / dt, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ("bf16 storage + f32 accumulate (V2V), " + args.backbone_dtype + " backbone") if bf16 else
                 ("f32" if args.backbone_dtype == "fp32" else "f32 V2V, bf16 backbone"), "data": "synthetic",
        "config": {"workload": f"batch={args.batch}/GPU synthetic 256x256 image N(0,1) + {args.depth_kind} depth 1024x1280, "
                               f"{G}^3 grid, 15 joints, " + ("bf16 V2V (BASELINE configs[2])" if bf16 else "fp32 (BASELINE configs[1])"),
                   "batch_per_gpu": args.batch, "global_batch": args.batch * world, "volume_size": G,
                   "parallelism": f"dp{world}" + (" + RCCL all_gather of [B,15,3] joints" if world > 1 else ""),
                   "hipgraph": bool(args.graphs)},
    }
    # ---- roofline of the dominant kernel, from the HIP events of the timed region ----------------
    key = ("conv3d", 3, 32, 32, G)
    if key in prof:
        ms = prof[key]
        avg_ms = sum(ms) / len(ms)
        flop = 2.0 * args.batch * G ** 3 * 27 * 32 * 32          # algorithmic FLOP of one launch
        ach = flop / (avg_ms * 1e-3) / 1e12
        conv_ms_per_step = sum(sum(v) for v in prof.values()) / args.steps
        k7 = [v for k, v in prof.items() if k[1] == 7]
        stage_tflops = V2V_GFLOP_PER_FRAME.get(G, 0) * args.batch / (conv_ms_per_step * 1e-3) / 1e3 if conv_ms_per_step else None
        line["roofline"] = {
            "bound": "mfma", "achieved": round(ach, 2), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / F32_MFMA_PEAK_TFLOPS, 4), "traffic": pmc_traffic(args.batch, G),
            "kernel": f"conv3d 3x3x3 32->32 @{G}^3 f32 (conv3d_k3_wino43pp_kernel: 1-D Winograd F(4,3) on v_mfma_f32_16x16x4_f32, ping-pong wave groups), "
                      f"{len(ms) // args.steps} launches/step",
            "note": "achieved = ALGORITHMIC (direct-convolution) FLOP / launch time; the kernel executes 1/2 of them on the "
                    "matrix cores (Winograd F(4,3) along z), so executed_mfma_frac = frac / 2 is the pipe utilisation",
            "executed_mfma_frac": round(ach * 0.5 / F32_MFMA_PEAK_TFLOPS, 4),
            "avg_launch_ms": round(avg_ms, 4), "flop_per_launch": flop,
            "stage": {"v2v_conv_ms_per_step": round(conv_ms_per_step, 3),
                      "v2v_tflops": round(stage_tflops, 2) if stage_tflops else None,
                      "v2v_hbm_frac": round(V2V_GB_PER_FRAME.get(G, 0) * args.batch / (conv_ms_per_step * 1e-3) / HBM_PEAK_GBS, 4)
                      if conv_ms_per_step else None,
                      "conv7_avg_ms": round(sum(k7[0]) / len(k7[0]), 4) if k7 else None},
        }
    if args.dump_kernel_events:
        for k in sorted(prof, key=lambda k: -sum(prof[k])):
            v = prof[k]
            print(f"{str(k):44s} {len(v) // args.steps:3d}/step avg {sum(v) / len(v):8.4f} ms  per-step {sum(v) / args.steps:8.4f} ms",
                  file=sys.stderr)
    keyb = ("conv3d_bf16", 3, 32, 32, G)
    if keyb in prof:
        ms = prof[keyb]
        avg_ms = sum(ms) / len(ms)
        nbytes = 2.0 * args.batch * G ** 3 * (32 + 32)           # bf16 input + output records of one launch (skip reads excluded)
        flop = 2.0 * args.batch * G ** 3 * 27 * 32 * 32
        conv_ms_per_step = sum(sum(v) for v in prof.values()) / args.steps
        k7 = [v for k, v in prof.items() if k[1] == 7]
        line["roofline"] = {
            "bound": "hbm", "achieved": round(nbytes / (avg_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(nbytes / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
            "kernel": f"conv3d 3x3x3 32->32 @{G}^3 bf16 storage (v_mfma_f32_16x16x32_bf16), {len(ms) // args.steps} launches/step",
            "avg_launch_ms": round(avg_ms, 4), "bytes_per_launch": nbytes,
            "mfma_tflops": round(flop / (avg_ms * 1e-3) / 1e12, 1), "mfma_frac_of_2500": round(flop / (avg_ms * 1e-3) / 2.5e15, 4),
            "stage": {"v2v_conv_ms_per_step": round(conv_ms_per_step, 3),
                      "v2v_hbm_frac": round(0.5 * V2V_GB_PER_FRAME.get(G, 0) * args.batch / (conv_ms_per_step * 1e-3) / HBM_PEAK_GBS, 4),
                      "conv7_avg_ms": round(sum(k7[0]) / len(k7[0]), 4) if k7 else None},
        }
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(sd, G)
    if world == 1 and not bf16 and args.backbone_dtype == "fp32" and G == 64 and not args.no_extras:
        line["extra"] = {"config3_bf16_b32": config3_extra(net, rank, device, args.depth_kind)}
    print(json.dumps(line))


def config3_extra(net, rank, device, depth_kind):
    """BASELINE configs[2] measured beside the headline (never part of `value`): batch 32, bf16-storage V2V + bf16 backbone."""
    try:
        img, depth = device_inputs(32, rank, device, depth_kind)
        net.set_v2v_dtype("bf16")
        net.set_backbone_dtype("bf16")
        with torch.no_grad():
            for _ in range(2):
                net(img, net.grid_coord_proj_batch, net.coord_volumes, depth_map_batch=depth)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                net(img, net.grid_coord_proj_batch, net.coord_volumes, depth_map_batch=depth)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 5
        return {"value": round(32 / dt, 1), "unit": "frames/s", "ms_per_step": round(dt * 1e3, 3), "batch": 32,
                "dtype": "bf16 storage + f32 accumulate (V2V), bf16 backbone",
                "note": "joint error vs the float32 reference ~1e-2 m (> the 1e-3 parity tolerance): reported separately, see DESIGN.md 4b"}
    except Exception as e:      # never let the side measurement break the headline line
        return {"error": repr(e)[:200]}
    finally:
        net.set_v2v_dtype("fp32")
        net.set_backbone_dtype("fp32")


if __name__ == "__main__":
    main()
