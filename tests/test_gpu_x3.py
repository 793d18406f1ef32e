"""hip_dtype = 'bf16x3': the exact path's convolutions at fp32 grade on the bf16 matrix cores (drk_conv_fwd_x3: three
bf16 planes per operand, six partial products, fp32 accumulation).  Bars: the kernel against a float64 convolution of
the same fp32 operands, no looser than 4x the exact-fp32 kernel's own error on the same case and <= 4e-6 of the output
scale; logits against the exact-fp32 path <= 2e-5 and against fixture F2 within the fp32 bar; tokens of `_sample` EQUAL
to the reference's on fixtures F4 (full, short, time-conditioned)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import drakes_amd
from conftest import load_golden
from drakes_amd import ops

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('B,L,d,taps', [(3, 200, 1, 9), (2, 77, 4, 9), (2, 200, 16, 9), (3, 200, 64, 9), (1, 1024, 64, 9),
                                        (5, 33, 0, 1), (1, 1, 1, 9), (2, 129, 16, 9),
                                        (128, 200, 4, 9), (131, 200, 1, 9), (129, 199, 0, 1),    # 128-row tiles (few tiles)
                                        (131, 199, 16, 9), (3, 77, 16, 9), (1, 20, 16, 9), (300, 200, 16, 9)])   # 64-row tiles, 64-row halo
def test_conv_x3_kernel(dev, B, L, d, taps):
    g = torch.Generator(device='cpu').manual_seed(B * 1000 + L + d)
    w = torch.randn(128, 128, taps, generator=g) * 0.05
    h = torch.randn(B, L, 128, generator=g) * torch.linspace(0.2, 3.0, 128)      # channels of different scale
    bias = torch.randn(128, generator=g) * 0.1
    resid = torch.randn(B, L, 128, generator=g)
    dd = max(d, 1)
    pre = F.conv1d(h.double().permute(0, 2, 1), w.double(), bias.double(), dilation=dd, padding=(taps // 2) * dd).permute(0, 2, 1)
    want = torch.relu(pre) + resid.double()
    hd, rd, bd = h.to(dev), resid.to(dev), bias.to(dev)
    m3 = torch.empty(B, L, 128, dtype=torch.uint8, device=dev)
    o1 = ops.conv_fwd_f32(hd, ops.pack_conv_weight_f32(w.to(dev)), bd, rd, d, relu=True)
    o3 = ops.conv_fwd_x3(hd, ops.pack_conv_weight_x3(w.to(dev)), bd, rd, d, relu=True, relu_mask=m3)
    scale = float(pre.abs().max())
    e1 = float((o1.double().cpu() - want).abs().max())
    e3 = float((o3.double().cpu() - want).abs().max())
    print(f'max error: exact-fp32 kernel {e1:.2e}, three-plane kernel {e3:.2e}, scale {scale:.1f}')
    assert e3 <= 4e-6 * scale and e3 <= 4 * e1 + 1e-6 * scale
    clear = pre.abs() > 1e-4 * scale
    assert torch.equal(m3.cpu().bool()[clear], (pre > 0)[clear])
    # no ReLU, no bias, no residual (the input-gradient use), in place on the residual otherwise
    o = ops.conv_fwd_x3(hd, ops.pack_conv_weight_x3(w.to(dev)), None, None, d, relu=False)
    want2 = pre - bias.double()
    assert float((o.double().cpu() - want2).abs().max()) <= 4e-6 * scale
    ops.conv_fwd_x3(hd, ops.pack_conv_weight_x3(w.to(dev)), bd, rd, d, relu=True, out=rd)
    assert torch.equal(rd, o3)


def test_x3_logits_vs_exact_and_golden(gpu_model, dev):
    g = load_golden('f2_backbone.npz')
    x = torch.from_numpy(g['x_int']).to(dev)
    soft = torch.from_numpy(g['soft']).to(dev)
    sigma = torch.from_numpy(g['sigma']).to(dev)
    bb = gpu_model.backbone
    with torch.no_grad():
        for inp, key in ((x, 'logits_int_tc'), (soft, 'logits_soft_tc')):
            tb = bb.time_bias(sigma)
            exact = bb.forward_f32(inp, tb, tb.stride(0)).clone()
            got = bb.forward_x3(inp, tb, tb.stride(0))
            d = float((got - exact).abs().max())
            print('max |logit(x3) - logit(exact fp32)|:', d)
            assert d <= 2e-5
            np.testing.assert_allclose(got.cpu().numpy(), g[key], rtol=1e-4, atol=2e-4)


def test_x3_sample_tokens_equal_the_fixtures(dev, det_sd, gpu_model):
    g = load_golden('f4_sample.npz')
    try:
        gpu_model.hip_dtype = 'bf16x3'
        toks = gpu_model._sample(eval_sp_size=8, seed=int(g['useed'])).cpu()
        assert (toks != torch.from_numpy(g['tokens'])).float().mean().item() == 0.0
        gpu_model.hip_graph = True
        assert torch.equal(gpu_model._sample(eval_sp_size=8, seed=int(g['useed'])).cpu(), toks)
    finally:
        gpu_model.hip_dtype = 'fp32'
        gpu_model.hip_graph = False
    for name, kw in (('f4_sample_short.npz', dict(length=64, steps=16)), ('f4_sample_tc.npz', dict(time_conditioning=True))):
        g = load_golden(name)
        m = drakes_amd.Diffusion(drakes_amd.default_config(**kw)).to(dev)
        m.load_state_dict(det_sd)
        m.hip_dtype = 'bf16x3'
        want = torch.from_numpy(g['tokens'])
        toks = m._sample(eval_sp_size=want.shape[0], seed=int(g['useed'])).cpu()
        assert (toks != want).float().mean().item() == 0.0, name


def test_x3_gradients_vs_oracle(dev, det_sd):
    """hip_soft_dtype = 'bf16x3': forward and input-gradient convolutions on drk_conv_fwd_x3, weight gradients exact fp32:
    parameter gradients against the oracle's autograd at the fp32 bar (2e-3 of each tensor's max)."""
    from oracle import drakes_oracle as O
    from test_gpu_finetune import _check_param_grads, _input_away_from_relu_kinks, _oracle_grads
    B, L = 2, 45
    m = drakes_amd.Diffusion(drakes_amd.default_config(length=L, time_conditioning=True)).to(dev)
    m.load_state_dict(det_sd)
    sigma = torch.tensor([0.3, 4.0])
    x, gen = _input_away_from_relu_kinks(det_sd, B, L, True, sigma, 5)
    Gm = torch.from_numpy(gen.standard_normal((B, L, 5)).astype(np.float32))
    xo = x.clone().requires_grad_()
    sdg = _oracle_grads(det_sd, lambda sd: (O.cnn_forward(sd, xo, sigma) * Gm).sum())
    xg = x.to(dev).requires_grad_()
    m.backbone.train_dtype = 'bf16x3'
    (m.backbone(xg, sigma.to(dev)) * Gm.to(dev)).sum().backward()
    print('worst parameter gradient error (bf16x3):', _check_param_grads(m, sdg))
    assert float((xg.grad.cpu() - xo.grad).abs().max()) / float(xo.grad.abs().max()) < 2e-3


def test_x3_sample_finetune_gradient_golden(dev, det_sd):
    """Fixture F5 (16 steps, K = 6) with hip_dtype = 'bf16x3': outputs and the six gradient tensors at the fp32 path's bars."""
    g = load_golden('f5_finetune_s16k6.npz')
    steps, trunc, temp = int(g['steps']), int(g['trunc']), float(g['temp'])
    m = drakes_amd.Diffusion(drakes_amd.default_config(steps=steps)).to(dev)
    m.load_state_dict(det_sd)
    m.train()
    m.hip_dtype = 'bf16x3'
    m.config.finetuning.truncate_steps = trunc
    m.config.finetuning.gumbel_softmax_temp = temp
    out, lx, ct, mct, cf = m._sample_finetune_gradient(eval_sp_size=2, seed=int(g['useed']))
    np.testing.assert_allclose(out.detach().cpu().numpy(), g['sample'], rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(lx[-1].detach().cpu().numpy(), g['last_x_final'], rtol=1e-4, atol=2e-4)
    rw = torch.from_numpy(g['reward_w']).to(dev)
    (-(out.permute(0, 2, 1) * rw).sum((1, 2)).mean()).backward()
    params = dict(m.named_parameters())
    for key in g.files:
        if key.startswith('grad_backbone'):
            ref = torch.from_numpy(g[key])
            got = params[key[len('grad_'):]].grad.cpu()[:ref.shape[0]]
            e = (got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-12)
            assert e < 5e-3, (key, e)
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters() if p.grad is not None)).item()
    assert abs(gn - float(g['grad_norm'])) <= 2e-3 * float(g['grad_norm'])
