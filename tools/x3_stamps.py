"""Phase times of drk_conv_fwd_x3 (second tile of every workgroup) from a -DX3_STAMPS build:
    python tools/exp.py build x3s -DX3_STAMPS;  DRK_LIB=$PWD/drakes_amd/libexp_x3s.so python tools/x3_stamps.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from drakes_amd import ops  # noqa: E402
dev = torch.device('cuda:0')
B, L = 1024, 200
g = torch.Generator(device='cpu').manual_seed(0)
w3 = ops.pack_conv_weight_x3((torch.randn(128, 128, 9, generator=g) * 0.05).to(dev))
h = torch.randn(B, L, 128, device=dev)
feat = torch.randn(B, L, 128, device=dev)
bias = torch.zeros(128, device=dev)
mask = torch.zeros(B, L, 128, dtype=torch.uint8, device=dev)
for d in (1, 16):
    for _ in range(3):
        ops.conv_fwd_x3(h, w3, bias, feat, d, relu=True, out=feat, relu_mask=mask)
    torch.cuda.synchronize()
    s = mask.view(-1)[: 256 * 16 * 8].view(torch.int64).reshape(256, 16).cpu().numpy().astype(np.float64) * 0.01
    s = s[s[:, 0] > 0]
    names = ['tap 0 MFMAs', 'barrier', 'tap 1 MFMAs', 'barrier', 'tap 2 MFMAs', 'barrier', 'tap 3 MFMAs', 'barrier',
             'taps 4.. (last stamped: tap 8)', 'barrier', 'exchange', 'epilogue']
    print(f'dilation {d}: {len(s)} workgroups; tile total {np.mean(s[:, 12] - s[:, 0]):.2f} us')
    idx = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12]
    for k, nme in enumerate(names):
        dt = s[:, idx[k + 1]] - s[:, idx[k]]
        print(f'   {nme:32s} mean {dt.mean():6.2f}  p10 {np.percentile(dt, 10):6.2f}  p90 {np.percentile(dt, 90):6.2f} us')
