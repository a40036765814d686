import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from sceneego_amd import pose_resnet
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
net = pose_resnet.get_pose_net(None).to("cuda:0").eval()
for B in (8, 32):
    x = torch.randn(B, 3, 256, 256, device="cuda:0")
    with torch.no_grad():
        for dt in (torch.float32, torch.bfloat16):
            fb = pose_resnet.FoldedBackbone(net, dtype=dt)
            torch.backends.cudnn.benchmark = False
            t0 = timeit(lambda: fb(x))
            torch.backends.cudnn.benchmark = True
            t1 = timeit(lambda: fb(x))
            torch.backends.cudnn.benchmark = False
            print(f"B={B} {dt}: benchmark=False {t0:.3f} ms, benchmark=True {t1:.3f} ms")
