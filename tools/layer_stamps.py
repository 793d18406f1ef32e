"""Phase timeline of drk_layer_bf16 from a -DLT_STAMPS build (tools/exp.py build lst -DLT_STAMPS), not part of the product.

    DRK_LIB=$PWD/drakes_amd/libexp_lst.so python tools/layer_stamps.py [--batch 256] [--length 1024]

Stamps per workgroup (100 MHz wall clock, thread 0): 0 start, 1 h tile parked (HALO builds), 2 tap loop issued, 3 residual
stores issued, 4 LayerNorm partials exchanged, 5 h' stores issued, 6 all memory operations acknowledged, 7 = XCC / HW id.
"""
import argparse, ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import drakes_amd  # noqa: E402
from drakes_amd import _lib, fastpath, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=256)
ap.add_argument('--length', type=int, default=1024)
a = ap.parse_args()
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = drakes_amd.Diffusion(drakes_amd.default_config(length=a.length)).to(dev).eval()
bb = m.backbone
st = fastpath._packed(bb)
B, L = a.batch, a.length
feat = torch.randn(B, L, 128, device=dev)
h0 = torch.randn(B, L, 128, device=dev).to(torch.bfloat16)
h1 = torch.empty_like(h0)
tbias = bb.time_bias(torch.zeros(1, device=dev))
vec = st['vecs']
P = lambda t: ctypes.c_void_p(t.data_ptr())
nwg_max = (B * L + 63) // 64
mean = torch.zeros(max(B * L, nwg_max * 16), device=dev)
rstd = torch.zeros_like(mean)
for i, d in ((0, 1), (2, 4), (3, 16), (4, 64)):
    nwg = (B * L + 127) // 128
    args = (P(h0), P(feat), P(feat), P(h1), P(st['wstream'][9 * i:]), P(vec[3 * (i + 1)]), P(tbias[0, i + 1]), 0,
            P(vec[3 * (i + 1) + 1]), P(vec[3 * (i + 1) + 2]), B, L, 9, d, 1, 1, None, P(mean), P(rstd), ops._stream())
    for _ in range(3):
        _lib.call('drk_layer_bf16', *args)
    torch.cuda.synchronize()
    s = mean.view(torch.int64)[: nwg * 8].reshape(nwg, 8).cpu().numpy()
    t = (s[:, :7] - s[:, 0].min()).astype(np.float64) * 0.01           # us
    c = rstd.view(torch.int64)[: nwg * 8].reshape(nwg, 8).cpu().numpy().astype(np.float64)
    tapc = c[:, 2] - c[:, 1 if d != 64 else 0]
    tapt = (s[:, 2] - s[:, 1 if d != 64 else 0]).astype(np.float64) * 0.01
    print(f'   tap loop: {np.median(tapc):.0f} shader cycles median, clock {np.median(tapc / np.maximum(tapt, 0.01)) / 1e3:.3f} GHz; '
          f'whole workgroup clock {np.median((c[:, 6] - c[:, 0]) / ((s[:, 6] - s[:, 0]) * 0.01)) / 1e3:.3f} GHz')
    if d == 64:
        t[:, 1] = t[:, 0]
    names = ['load h + park', 'tap loop', 'bias/relu/residual + stores issued', 'LN partials + barrier', 'LN apply + h\' out', 'drain']
    dur = np.diff(t, axis=1)
    print(f'dilation {d}: launch span {t[:, 6].max():.1f} us; workgroup residency mean {np.mean(t[:, 6] - t[:, 0]):.1f} us')
    for k, nme in enumerate(names):
        print(f'   {nme:40s} mean {dur[:, k].mean():6.2f}  p10 {np.percentile(dur[:, k], 10):6.2f}  p90 {np.percentile(dur[:, k], 90):6.2f} us')
    # concurrency: how many workgroups are inside each phase over time
    grid_t = np.arange(0, t[:, 6].max(), 1.0)
    occ = np.zeros((len(grid_t), 6))
    for k in range(6):
        occ[:, k] = ((t[:, k][None, :] <= grid_t[:, None]) & (grid_t[:, None] < t[:, k + 1][None, :])).sum(1)
    print('   t(us): workgroups in [load, taps, resid, lnbar, lnout, drain]')
    for j in range(0, len(grid_t), max(1, len(grid_t) // 24)):
        print(f'   {grid_t[j]:6.0f}: ' + ' '.join(f'{int(v):4d}' for v in occ[j]))
    hw = s[:, 7]
    cu = (hw & 0xffffffff)
    xcc = hw >> 32
    # HW_ID: wave_id[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13] ...
    cuid = ((cu >> 8) & 0xf) | (((cu >> 12) & 1) << 4) | (((cu >> 13) & 7) << 5) | ((xcc & 0xf) << 8)
    uniq = np.unique(cuid)
    first = np.argsort(t[:, 0])[:512]
    print(f'   distinct CUs seen: {len(uniq)}; first 512 workgroups started within {t[first, 0].max():.2f} us; '
          f'start order vs blockIdx monotone: {np.all(np.diff(t[np.arange(512), 0]) >= -0.5)}')
    print('   CU of workgroups 0..31:', cuid[:32].tolist())
    same = [int(np.flatnonzero(cuid[:512] == cuid[i])[1]) if (cuid[:512] == cuid[i]).sum() > 1 else -1 for i in range(8)]
    print('   co-resident partner (blockIdx) of workgroups 0..7 among the first 512:', same)
    starts = np.sort(t[:, 0])
    print('   start-time deciles:', np.percentile(starts, [0, 10, 25, 50, 75, 90, 100]).round(1))
