"""Backbone epilogue A/B: MIOpen convolution without bias + se_bias_act_nchw_f32 (production) against torch's fused MIOpen entry points
(aten::miopen_convolution_relu / miopen_convolution_add_relu: miopenConvolutionBiasActivationForward where a fusion plan exists).
usage: python tools/diag/backbone_fused_ops.py [batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from bench import build_network, device_inputs
from sceneego_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
net, _ = build_network(64, dev)
img, depth = device_inputs(B, 0, dev, "uniform")
with torch.no_grad():
    net(img, net.grid_coord_proj_batch, net.coord_volumes, depth_map_batch=depth)
fb = net._folded[0]
ba = _lib.bias_act_nchw
mrelu = torch.ops.aten.miopen_convolution_relu
maddrelu = torch.ops.aten.miopen_convolution_add_relu


def fused(images):
    x = mrelu(images.contiguous(), fb.stem[0], fb.stem[1], [2, 2], [3, 3], [1, 1], 1)
    x = F.max_pool2d(x, 3, stride=2, padding=1)
    for c1, c2, c3, stride, ds in fb.blocks:
        y = mrelu(x, c1[0], c1[1], [1, 1], [0, 0], [1, 1], 1)
        y = mrelu(y, c2[0], c2[1], list(stride), [1, 1], [1, 1], 1)
        sc = x if ds is None else F.conv2d(x, ds[0], ds[1], stride=ds[2])
        x = maddrelu(y, c3[0], sc, 1.0, c3[1], [1, 1], [0, 0], [1, 1], 1)
    for li, (w, b) in enumerate(fb.ups):
        Bn, _, H, W = x.shape
        if Bn * H * W <= fb.DECONV_GEMM_MAX_POSITIONS:
            x = fb._deconv_gemm(li, x, b)
        else:
            x = ba(F.conv_transpose2d(x, w, None, stride=2, padding=1), b, None, True)
    return x


def timeit(fn, n=20):
    with torch.no_grad():
        for _ in range(3):
            out = fn(img)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            out = fn(img)
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, out


for r in range(2):
    ta, oa = timeit(lambda im: fb(im))
    tb, ob = timeit(fused)
    print(f"B={B} round {r}: production (conv + se_bias_act) {ta:.4f} ms | miopen_convolution_relu / add_relu {tb:.4f} ms | max|diff| {float((oa - ob).abs().max()):.2e} of {float(oa.abs().max()):.2f}")
