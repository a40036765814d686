import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from sceneego_amd import pose_resnet
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
net = pose_resnet.get_pose_net(None).to("cuda:0").eval()
for B in (8, 32):
    x = torch.randn(B, 3, 256, 256, device="cuda:0")
    with torch.no_grad():
        f32 = pose_resnet.FoldedBackbone(net)
        ref = f32(x)
        print(f"B={B} fp32 NCHW fused epilogue : {timeit(lambda: f32(x)):.3f} ms")
        for cl in (False, True):
            fb = pose_resnet.FoldedBackbone(net, dtype=torch.bfloat16, channels_last=cl)
            y = fb(x).float()
            print(f"B={B} bf16 channels_last={cl}: {timeit(lambda: fb(x)):.3f} ms  rel err {float((y - ref).abs().max() / ref.abs().max()):.3e}")
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                fb(x); fb(x)
            torch.cuda.current_stream().wait_stream(s)
            with torch.cuda.graph(g):
                y = fb(x)
            print(f"    graph replay: {timeit(lambda: g.replay()):.3f} ms")
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = f32(x)
        print(f"B={B} fp32 graph replay: {timeit(lambda: g.replay()):.3f} ms")
