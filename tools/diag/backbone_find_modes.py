"""MIOpen solver selection for the 2-D backbone: the folded ResNet-50 + deconv head at B=8 under several
MIOpen find configurations, each in its own process (the find mode is read when the library initialises).

  python tools/diag/backbone_find_modes.py [--batch 8] [--modes default,bench,normal,search] [--db gpurun_out/miopen_db]

Per mode: device time of the whole backbone (HIP events, 20 runs) and of every distinct convolution shape on its own; with
--db the user find-db / perf-db a mode wrote is kept (MIOPEN_USER_DB_PATH) so that a later process - or the product, through
sceneego_amd/_lib.py - can start from it.  Output: one table per mode + a JSON line per mode.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

MODES = {
    # name: (env, cudnn.benchmark)
    "default": ({}, False),                                       # immediate mode: find-db / heuristics, no search
    "bench": ({}, True),                                          # torch.backends.cudnn.benchmark: miopenFind* with the default find mode
    "normal": ({"MIOPEN_FIND_MODE": "1"}, True),                  # full Find: every applicable solver is timed
    "search": ({"MIOPEN_FIND_MODE": "1", "MIOPEN_FIND_ENFORCE": "3"}, True),       # + tuning of the tunable solvers (slow)
    "fast": ({"MIOPEN_FIND_MODE": "2"}, True),
    "hybrid": ({"MIOPEN_FIND_MODE": "3"}, True),
}


def child(args):
    import torch
    import torch.nn.functional as F
    sys.path.insert(0, ROOT)
    from sceneego_amd import _lib, pose_resnet
    torch.backends.cudnn.benchmark = bool(int(os.environ.get("SE_BENCHMARK", "0")))
    dev = "cuda:0"
    _lib.load()
    net = pose_resnet.get_pose_net(None).eval()
    net = net.to(dev)
    fb = pose_resnet.FoldedBackbone(net)
    B = args.batch
    x = torch.randn(B, 3, 256, 256, device=dev)

    def ev_time(fn, n=20, warm=3):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        ts.sort()
        return ts[len(ts) // 2]

    t0 = time.perf_counter()
    with torch.no_grad():
        fb(x)                      # solver selection / search happens here
        torch.cuda.synchronize()
        t_first = time.perf_counter() - t0
        whole = ev_time(lambda: fb(x))
        # the distinct convolution shapes, one by one
        shapes = []
        seen = set()

        def rec(kind, xin, w, stride, pad):
            key = (kind, tuple(xin.shape), tuple(w.shape), stride, pad)
            if key not in seen:
                seen.add(key)
                shapes.append((kind, xin.clone(), w, stride, pad))

        h = F.conv2d(x, fb.stem[0], None, stride=2, padding=3); rec("conv", x, fb.stem[0], 2, 3)
        h = F.max_pool2d(torch.relu(h), 3, stride=2, padding=1)
        for c1, c2, c3, stride, ds in fb.blocks:
            st = stride[0] if isinstance(stride, tuple) else stride
            rec("conv", h, c1[0], 1, 0); y = torch.relu(F.conv2d(h, c1[0]))
            rec("conv", y, c2[0], st, 1); y = torch.relu(F.conv2d(y, c2[0], None, stride=st, padding=1))
            rec("conv", y, c3[0], 1, 0); y = F.conv2d(y, c3[0])
            if ds is not None:
                dst = ds[2][0] if isinstance(ds[2], tuple) else ds[2]
                rec("conv", h, ds[0], dst, 0); h = F.conv2d(h, ds[0], None, stride=dst)
            h = torch.relu(y + h)
        for w, b in fb.ups:
            rec("deconv", h, w, 2, 1); h = torch.relu(F.conv_transpose2d(h, w, None, stride=2, padding=1))
        rows = []
        total = 0.0
        for kind, xin, w, stride, pad in shapes:
            if kind == "conv":
                fn = lambda: F.conv2d(xin, w, None, stride=stride, padding=pad)
                co, ci, kh = w.shape[0], w.shape[1], w.shape[2]
                ho = (xin.shape[2] + 2 * pad - kh) // stride + 1
                flop = 2.0 * B * ho * ho * co * ci * kh * kh
            else:
                fn = lambda: F.conv_transpose2d(xin, w, None, stride=2, padding=1)
                ci, co, kh = w.shape[0], w.shape[1], w.shape[2]
                ho = xin.shape[2] * 2
                flop = 2.0 * B * xin.shape[2] ** 2 * co * ci * kh * kh
            ms = ev_time(fn, n=10, warm=2)
            rows.append({"kind": kind, "in": list(xin.shape), "w": list(w.shape), "stride": stride, "us": round(ms * 1e3, 1),
                         "tflops": round(flop / (ms * 1e-3) / 1e12, 1)})
        for r in rows:
            print(f"  {r['kind']:6s} in {str(r['in']):22s} w {str(r['w']):20s} s{r['stride']}  {r['us']:8.1f} us  {r['tflops']:6.1f} TFLOP/s")
    print(json.dumps({"mode": os.environ.get("SE_MODE"), "batch": B, "backbone_ms": round(whole, 4), "first_call_s": round(t_first, 1),
                      "conv_rows": rows}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--modes", default="default,bench,normal,search")
    ap.add_argument("--db", default="", help="directory for MIOPEN_USER_DB_PATH (kept; one sub-directory per mode)")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--timeout", type=int, default=1500)
    args = ap.parse_args()
    if args.child:
        return child(args)
    for m in args.modes.split(","):
        env_add, bench = MODES[m]
        env = dict(os.environ, SE_MODE=m, SE_BENCHMARK="1" if bench else "0", **env_add)
        if args.db:
            d = os.path.abspath(os.path.join(args.db, m))
            os.makedirs(d, exist_ok=True)
            env["MIOPEN_USER_DB_PATH"] = d
            env["MIOPEN_CUSTOM_CACHE_DIR"] = os.path.join(d, "cache")
        print(f"=== mode {m}: env {env_add} cudnn.benchmark={bench}", flush=True)
        t0 = time.time()
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--batch", str(args.batch)], env=env,
                               timeout=args.timeout)
            print(f"=== mode {m}: rc {r.returncode}, {time.time() - t0:.0f} s", flush=True)
        except subprocess.TimeoutExpired:
            print(f"=== mode {m}: TIMEOUT after {args.timeout} s", flush=True)


if __name__ == "__main__":
    main()
