"""Time the folded 2D backbone (MIOpen) variants on the GPU."""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sceneego_amd import pose_resnet  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = "cuda:0"
    net = pose_resnet.get_pose_net(None).to(dev).eval()
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    x = torch.randn(B, 3, 256, 256, device=dev)
    fb = pose_resnet.FoldedBackbone(net, channels_last=True)
    with torch.no_grad():
        print(f"B={B} folded fp32 channels_last, benchmark=False: {timeit(lambda: fb(x)):.3f} ms")
        torch.backends.cudnn.benchmark = True
        print(f"B={B} folded fp32 channels_last, benchmark=True : {timeit(lambda: fb(x)):.3f} ms")
        xs = x.contiguous()
        torch.backends.cudnn.benchmark = False
        print(f"B={B} eager module (NCHW, unfolded)             : {timeit(lambda: net(xs, compute_heatmaps=False)):.3f} ms")
        fb16 = pose_resnet.FoldedBackbone(net, dtype=torch.bfloat16, channels_last=True)
        print(f"B={B} folded bf16 channels_last                  : {timeit(lambda: fb16(x)):.3f} ms")
        torch.backends.cudnn.benchmark = True
        print(f"B={B} folded bf16 channels_last benchmark=True   : {timeit(lambda: fb16(x)):.3f} ms")
        torch.backends.cudnn.benchmark = False

        class NCHW(pose_resnet.FoldedBackbone):
            def __init__(self, net, fused):
                super().__init__(net)
                self.fused = fused
                c = lambda wb: (wb[0].contiguous(), wb[1].contiguous())
                self.stem = c(self.stem)
                self.blocks = [(c(a), c(b), c(d), st, None if ds is None else c(ds[:2]) + (ds[2],)) for a, b, d, st, ds in self.blocks]
                self.ups = [c(u) for u in self.ups]

            def conv_relu(self, x, w, b, stride=1, padding=0):
                if self.fused:
                    return torch.miopen_convolution_relu(x, w, b, (stride, stride) if isinstance(stride, int) else stride,
                                                         (padding, padding), (1, 1), 1)
                return F.relu_(F.conv2d(x, w, b, stride=stride, padding=padding))

            def __call__(self, images):
                x = images.contiguous()
                x = self.conv_relu(x, self.stem[0], self.stem[1], 2, 3)
                x = F.max_pool2d(x, 3, stride=2, padding=1)
                for c1, c2, c3, stride, ds in self.blocks:
                    y = self.conv_relu(x, c1[0], c1[1])
                    y = self.conv_relu(y, c2[0], c2[1], stride, 1)
                    sc = x if ds is None else F.conv2d(x, ds[0], ds[1], stride=ds[2])
                    if self.fused:
                        x = torch.miopen_convolution_add_relu(y, c3[0], sc, 1.0, c3[1], (1, 1), (0, 0), (1, 1), 1)
                    else:
                        y = F.conv2d(y, c3[0], c3[1])
                        x = F.relu_(y.add_(sc))
                for w, b in self.ups:
                    x = F.relu_(F.conv_transpose2d(x, w, b, stride=2, padding=1))
                return x

        for fused in (False, True):
            try:
                nb = NCHW(net, fused)
                ref = fb(x)
                got = nb(x)
                err = float((got - ref).abs().max())
                print(f"B={B} folded fp32 NCHW fused={fused}: {timeit(lambda: nb(x)):.3f} ms  (max diff vs channels_last {err:.2e})")
            except Exception as e:  # noqa: BLE001
                print("variant failed", fused, repr(e)[:300])
        has = hasattr(torch, "miopen_convolution_relu")
        print("torch.miopen_convolution_relu available:", has)


if __name__ == "__main__":
    main()
