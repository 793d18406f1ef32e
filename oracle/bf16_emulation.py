"""TEST INFRASTRUCTURE — CPU emulation of the bf16 fast path's rounding points.

The bf16 sampling kernel (drakes_amd/csrc/drk_sample_bf16.hip) keeps the residual stream and
every accumulation in fp32 and rounds to bf16 (round-to-nearest-even) exactly where an MFMA
operand is formed: all matrix weights, the LayerNorm output h, the residual stream fed to
final_conv.0 and its ReLU output fed to final_conv.2.  Emulating those roundings on top of the
pinned fp32 oracle (drakes_oracle.cnn_forward, models/dnaconv.py:74-110) gives a reference the
kernel must match to ~1e-3 instead of the ~3e-2 that separates bf16 from fp32 arithmetic, so
indexing mistakes cannot hide inside the precision gap.
"""
import torch
import torch.nn.functional as F

from . import drakes_oracle as O


def r16(t):
    return t.to(torch.bfloat16).to(torch.float32)


def cnn_forward_bf16(sd, seq, sigma):
    if not (seq.ndim > 2 and seq.shape[-1] == O.VOCAB):
        seq = F.one_hot(seq, num_classes=O.VOCAB).float()
    temb = O.time_embedding(sd, sigma)                       # fp32 (drk_time_embed_fwd / drk_time_bias_fwd)
    feat = F.relu(F.conv1d(r16(seq).permute(0, 2, 1), r16(sd['backbone.linear.weight']),
                           sd['backbone.linear.bias'], padding=4))
    n = O.n_layers_of(sd)
    for i in range(n):
        d = O.layer_dilation(i, n)
        tb = F.linear(temb, sd[f'backbone.time_layers.{i}.dense.weight'], sd[f'backbone.time_layers.{i}.dense.bias'])
        h = feat + tb[:, :, None]
        h = F.layer_norm(h.permute(0, 2, 1), (h.shape[1],), sd[f'backbone.norms.{i}.weight'],
                         sd[f'backbone.norms.{i}.bias'])
        h = r16(h)
        h = F.relu(F.conv1d(h.permute(0, 2, 1), r16(sd[f'backbone.convs.{i}.weight']),
                            sd[f'backbone.convs.{i}.bias'], dilation=d, padding=4 * d))
        feat = h + feat
    y = F.relu(F.conv1d(r16(feat), r16(sd['backbone.final_conv.0.weight']), sd['backbone.final_conv.0.bias']))
    y = F.conv1d(r16(y), r16(sd['backbone.final_conv.2.weight']), sd['backbone.final_conv.2.bias'])
    return y.permute(0, 2, 1)


# ---- the differentiable bf16 path (drakes_amd/autograd.py::_BackboneBf16Fn) ----------------------------------------------
class _RoundSTE(torch.autograd.Function):
    """bf16 rounding in the forward pass, identity in the backward pass: what mixed-precision training differentiates
    (fp32 master weights and activations, bf16 copies as MFMA operands)."""

    @staticmethod
    def forward(ctx, x):
        return r16(x)

    @staticmethod
    def backward(ctx, g):
        return g


def cnn_forward_train_bf16(sd, seq, sigma, masks=None, head_mask=None, stem_mask=None):
    """CNNModel.forward (models/dnaconv.py:74-110) with the rounding points of the bf16 TRAINING path: fp32 stem, bf16
    LayerNorm outputs and conv weights (20 layers and final_conv.0), bf16 copy of the stream into final_conv.0, exact-fp32
    final_conv.2.  Differentiable through torch autograd (straight-through rounding).

    masks / head_mask / stem_mask: optional ReLU gates ([B,L,128] 0/1 per layer, for final_conv.0, for the stem) to use INSTEAD of the sign of
    this forward's own pre-activations.  ReLU's derivative is discontinuous, so two correct implementations whose
    pre-activations differ by rounding take different gates on the few elements next to 0, and each flipped gate switches
    a whole gradient term; feeding the kernel's own gates makes the gradient comparison about the backward arithmetic."""
    ste = _RoundSTE.apply
    if not (seq.ndim > 2 and seq.shape[-1] == O.VOCAB):
        seq = F.one_hot(seq, num_classes=O.VOCAB).float()
    temb = O.time_embedding(sd, sigma)
    c0 = F.conv1d(seq.permute(0, 2, 1), sd['backbone.linear.weight'], sd['backbone.linear.bias'], padding=4)
    feat = F.relu(c0) if stem_mask is None else c0 * stem_mask.permute(0, 2, 1).to(c0.dtype)
    n = O.n_layers_of(sd)
    for i in range(n):
        d = O.layer_dilation(i, n)
        tb = F.linear(temb, sd[f'backbone.time_layers.{i}.dense.weight'], sd[f'backbone.time_layers.{i}.dense.bias'])
        h = feat + tb[:, :, None]
        h = F.layer_norm(h.permute(0, 2, 1), (h.shape[1],), sd[f'backbone.norms.{i}.weight'], sd[f'backbone.norms.{i}.bias'])
        c = F.conv1d(ste(h).permute(0, 2, 1), ste(sd[f'backbone.convs.{i}.weight']), sd[f'backbone.convs.{i}.bias'],
                     dilation=d, padding=4 * d)
        a = F.relu(c) if masks is None else c * masks[i].permute(0, 2, 1).to(c.dtype)
        feat = a + feat
    c = F.conv1d(ste(feat), ste(sd['backbone.final_conv.0.weight']), sd['backbone.final_conv.0.bias'])
    y = F.relu(c) if head_mask is None else c * head_mask.permute(0, 2, 1).to(c.dtype)
    y = F.conv1d(y, sd['backbone.final_conv.2.weight'], sd['backbone.final_conv.2.bias'])
    return y.permute(0, 2, 1)
