"""CNN denoiser of DRAKES (DNA) on hand-written gfx950 kernels.

Same constructor, parameter names and state_dict keys as the reference `CNNModel`
(/root/reference/drakes_dna/models/dnaconv.py:37-110), so reference checkpoints load with
strict=True.  The torch modules below only HOLD parameters; `forward` runs the HIP kernels
(drakes_amd/csrc) on channels-last activations [B, L, 128].

Scope: the sampling configuration (clean_data=False, cls_free_guidance=False, classifier=False).  hidden_dim is 128
(the shipped checkpoints) or any smaller multiple of 8 (the reference takes it from args, dnaconv.py:46-63; cheap test
models): a narrower model keeps its own parameter shapes and state_dict, and runs zero-padded into the kernels' 128
lanes on the exact-fp32 kernels (LayerNorm takes its statistics over hidden_dim channels; padding lanes stay 0 in the
forward pass and carry no gradient).  The bf16 fast path is for hidden_dim = 128 only and says so.
"""
import copy

import torch
import torch.nn as nn

from .. import ops

DILATIONS = (1, 1, 4, 16, 64)   # dnaconv.py:53-57; padding = 4 * dilation ('same')


class GaussianFourierProjection(nn.Module):
    """Holds the fixed random frequencies W (dnaconv.py:8-21); evaluated inside drk_time_embed_fwd."""

    def __init__(self, embed_dim, scale=30.):
        super().__init__()
        self.W = nn.Parameter(torch.randn(embed_dim // 2) * scale, requires_grad=False)


class Dense(nn.Module):
    def __init__(self, input_dim, output_dim):
        super().__init__()
        self.dense = nn.Linear(input_dim, output_dim)


class _LazyPack(dict):
    """packed() entries plus some that are computed when first read (under no_grad: they are copies of the weights)."""

    def __init__(self, base, makers):
        super().__init__(base)
        self._makers = makers

    def __missing__(self, key):
        with torch.no_grad():
            value = self._makers[key]()
        self[key] = value
        return value


class CNNModel(nn.Module):
    def __init__(self, args, alphabet_size, num_cls, classifier=False):
        super().__init__()
        if classifier or getattr(args, 'clean_data', False) or getattr(args, 'cls_free_guidance', False):
            raise NotImplementedError('drakes_amd.CNNModel covers the sampling configuration only '
                                      '(clean_data=False, cls_free_guidance=False, classifier=False)')
        if not (8 <= args.hidden_dim <= ops.C and args.hidden_dim % 8 == 0):
            raise NotImplementedError(f'hidden_dim must be a multiple of 8 up to {ops.C} (the kernels\' lane count), '
                                      f'got {args.hidden_dim}')
        self.width = args.hidden_dim
        if alphabet_size != ops.V:
            raise NotImplementedError(f'alphabet_size must be {ops.V} (A,C,G,T,mask)')
        if getattr(args, 'dropout', 0.0) != 0.0:
            raise NotImplementedError('dropout > 0 is not implemented (the shipped config uses 0.0)')
        self.alphabet_size = alphabet_size
        self.args = args
        self.classifier = classifier
        self.num_cls = num_cls
        C = args.hidden_dim
        self.linear = nn.Conv1d(alphabet_size, C, kernel_size=9, padding=4)
        self.time_embedder = nn.Sequential(GaussianFourierProjection(embed_dim=C), nn.Linear(C, C))
        self.num_layers = 5 * args.num_cnn_stacks
        protos = [nn.Conv1d(C, C, kernel_size=9, dilation=d, padding=4 * d) for d in DILATIONS]
        # each layer type repeated num_cnn_stacks times consecutively, as deep copies (dnaconv.py:58)
        self.convs = nn.ModuleList([copy.deepcopy(p) for p in protos for _ in range(args.num_cnn_stacks)])
        self.time_layers = nn.ModuleList([Dense(C, C) for _ in range(self.num_layers)])
        self.norms = nn.ModuleList([nn.LayerNorm(C) for _ in range(self.num_layers)])
        self.final_conv = nn.Sequential(nn.Conv1d(C, C, kernel_size=1), nn.ReLU(),
                                        nn.Conv1d(C, alphabet_size, kernel_size=1))
        self.dropout = nn.Dropout(args.dropout)
        self._packed = None
        self._packed_key = None
        self._generation = 0            # bumped by invalidate(): part of every weight-cache key

    # ------------------------------------------------------------------ packed weights
    def dilation(self, i):
        return DILATIONS[i // self.args.num_cnn_stacks]

    def _weights_key(self):
        """Identity of the current weights for every derived cache (packed() layouts, the bf16 fragment stream,
        the sampler's schedule cache, the captured hipGraph, the fine-tune loop's logits reuse).  torch bumps
        `_version` on in-place tensor ops (optimizer steps, load_state_dict) but NOT on writes through `.data`
        (the reference's EMA helper does exactly that: models/ema.py:62,87 `param.data.copy_`): after such a write call
        invalidate() (or Diffusion.refresh_weights())."""
        return (self._generation,) + tuple((p.data_ptr(), p._version) for p in self.parameters())

    def invalidate(self):
        """Drops every kernel-layout copy of the weights; the next forward re-packs from the parameters."""
        self._generation += 1
        self._packed = None
        self._packed_key = None
        self._stream = None
        self._stream_key = None
        self._streamT = None
        self._streamT_key = None
        self._x3 = None
        self._x3_key = None

    # ------------------------------------------------------------------ hidden_dim < 128: the 128-lane image
    def wide(self, name, p):
        """Parameter `name` of a hidden_dim = width model as the same parameter of the 128-channel model that computes
        the same function on the first `width` channels: zero padding, except time_embedder.1.weight, whose input is
        [sin(width/2), cos(width/2)] (dnaconv.py:21) and lands on the 128-model's sin lanes 0.. and cos lanes 64.. ."""
        w, C = self.width, ops.C
        if w == C:
            return p
        pad = lambda t, *dims: torch.nn.functional.pad(t, [x for d in reversed(range(t.ndim)) for x in (0, (C - t.shape[d]) if d in dims else 0)])
        if name == 'time_embedder.0.W':
            return torch.nn.functional.pad(p, (0, C // 2 - w // 2))
        if name == 'time_embedder.1.weight':
            out = p.new_zeros(C, C)
            out[:w, :w // 2] = p[:, :w // 2]
            out[:w, C // 2:C // 2 + w // 2] = p[:, w // 2:]
            return out
        if name == 'final_conv.2.weight':
            return pad(p, 1)
        if name == 'final_conv.2.bias':
            return p
        if name == 'linear.weight':
            return pad(p, 0)
        if p.ndim == 1:
            return pad(p, 0)
        return pad(p, 0, 1)

    def narrow(self, name, g):
        """Inverse of wide() on a gradient in the 128-channel layout."""
        w, C = self.width, ops.C
        if w == C or g is None:
            return g
        if name == 'time_embedder.1.weight':
            return torch.cat([g[:w, :w // 2], g[:w, C // 2:C // 2 + w // 2]], dim=1)
        if name == 'final_conv.2.weight':
            return g[:, :w]
        if name == 'final_conv.2.bias':
            return g
        if name == 'linear.weight' or g.ndim == 1:
            return g[:w]
        return g[:w, :w]

    def packed(self):
        """Kernel-layout copies of the weights; rebuilt when any parameter changed in place
        (optimizer step, load_state_dict) or moved."""
        key = self._weights_key()
        if self._packed is None or key != self._packed_key:
            with torch.no_grad():
                n = self.num_layers
                P = {k: self.wide(k, v.detach()) for k, v in self.named_parameters()}
                # the exact-fp32 convolutions' layouts are built on first use: the bf16 paths, which come through here for
                # the stem / time / head weights after every optimizer step, never read them
                lazy = {
                    'conv_w': lambda: [ops.pack_conv_weight_f32(P[f'convs.{i}.weight']) for i in range(n)],
                    # input-gradient weights: the same kernel run on weight.transpose(0,1).flip(2)
                    'conv_wT': lambda: [ops.pack_conv_weight_f32(P[f'convs.{i}.weight'].transpose(0, 1).flip(2)) for i in range(n)],
                    'head0_wT': lambda: ops.pack_conv_weight_f32(P['final_conv.0.weight'].transpose(0, 1)),
                    'head0_w': lambda: ops.pack_conv_weight_f32(P['final_conv.0.weight']),
                }
                self._packed = _LazyPack({
                    'stem_w': ops.pack_stem_weight(P['linear.weight']),
                    'stem_b': P['linear.bias'].contiguous(),
                    'four_W': P['time_embedder.0.W'].contiguous(),
                    'temb_w': P['time_embedder.1.weight'].contiguous(),
                    'temb_b': P['time_embedder.1.bias'].contiguous(),
                    'dense_w': torch.stack([P[f'time_layers.{i}.dense.weight'] for i in range(n)]).contiguous(),
                    'dense_b': torch.stack([P[f'time_layers.{i}.dense.bias'] for i in range(n)]).contiguous(),
                    'ln_g': [P[f'norms.{i}.weight'].contiguous() for i in range(n)],
                    'ln_b': [P[f'norms.{i}.bias'].contiguous() for i in range(n)],
                    'conv_b': [P[f'convs.{i}.bias'].contiguous() for i in range(n)],
                    'head0_b': P['final_conv.0.bias'].contiguous(),
                    'head2_w': P['final_conv.2.weight'].reshape(self.alphabet_size, -1).contiguous(),
                    'head2_b': P['final_conv.2.bias'].contiguous(),
                }, lazy)
            self._packed_key = key
        return self._packed

    # ------------------------------------------------------------------ forward
    def time_bias(self, t):
        """[nb, n_layers, 128] per-layer time biases for sigma `t` [nb] (dnaconv.py:82,92)."""
        pk = self.packed()
        temb = ops.time_embed_fwd(t.contiguous(), pk['four_W'], pk['temb_w'], pk['temb_b'])
        return ops.time_bias_fwd(temb, pk['dense_w'], pk['dense_b'])

    def packed_x3(self):
        """The conv weights as three bf16 planes in fragment order (drk_conv_fwd_x3), cached like packed()."""
        key = self._weights_key()
        if getattr(self, '_x3', None) is None or key != self._x3_key:
            with torch.no_grad():
                P = {k: self.wide(k, v.detach()) for k, v in self.named_parameters()}
                n = self.num_layers
                self._x3 = {'conv_w3': [ops.pack_conv_weight_x3(P[f'convs.{i}.weight']) for i in range(n)],
                            'head0_w3': ops.pack_conv_weight_x3(P['final_conv.0.weight']),
                            # input-gradient weights (autograd): weight.transpose(0,1).flip(2)
                            'conv_w3T': [ops.pack_conv_weight_x3(P[f'convs.{i}.weight'].transpose(0, 1).flip(2)) for i in range(n)],
                            'head0_w3T': ops.pack_conv_weight_x3(P['final_conv.0.weight'].transpose(0, 1))}
            self._x3_key = key
        return self._x3

    def forward_x3(self, seq, tbias, tb_stride, work=None):
        """forward_f32 with the convolutions at fp32 grade on the bf16 matrix cores (three bf16 planes per operand,
        drk_conv_fwd_x3): logits within ~1e-5 of forward_f32's, about 3x its speed at large batches."""
        return self.forward_f32(seq, tbias, tb_stride, work, x3=True)

    def forward_f32(self, seq, tbias, tb_stride, work=None, x3=False):
        """Exact-fp32 kernels.  seq int64 [B,L] or float32 [B,L,5]; tbias [nb,n_layers,128]
        with `tb_stride` floats between batch rows (0 when one row serves the batch)."""
        pk = self.packed()
        B, L = seq.shape[0], seq.shape[1]
        if work is None:
            work = {}
        def buf(name, *shape):
            t = work.get(name)
            if t is None or t.shape != torch.Size(shape):
                t = torch.empty(*shape, dtype=torch.float32, device=seq.device)
                work[name] = t
            return t
        feat = ops.stem_fwd(seq, pk['stem_w'], pk['stem_b'], out=buf('feat', B, L, ops.C))
        h = buf('h', B, L, ops.C)
        for i in range(self.num_layers):
            ops.ln_fwd(feat, tbias[0, i], tb_stride, pk['ln_g'][i], pk['ln_b'][i], out=h, width=self.width)
            if x3:
                ops.conv_fwd_x3(h, self.packed_x3()['conv_w3'][i], pk['conv_b'][i], feat, self.dilation(i), relu=True, out=feat)
            else:
                ops.conv_fwd_f32(h, pk['conv_w'][i], pk['conv_b'][i], feat, self.dilation(i), relu=True, out=feat)
        if x3:
            ops.conv_fwd_x3(feat, self.packed_x3()['head0_w3'], pk['head0_b'], None, 0, relu=True, out=h)
        else:
            ops.conv_fwd_f32(feat, pk['head0_w'], pk['head0_b'], None, 0, relu=True, out=h)
        return ops.head_out_fwd(h, pk['head2_w'], pk['head2_b'], out=buf('logits', B, L, ops.V))

    def forward(self, seq, t, cls=None, return_embedding=False):
        """Reference signature (dnaconv.py:74).  Returns logits [B, L, 5] fp32."""
        if cls is not None or return_embedding:
            raise NotImplementedError('cls / return_embedding belong to the classifier and CFG variants')
        if seq.dtype != torch.int64 and not (seq.ndim > 2 and seq.shape[-1] == self.alphabet_size):
            seq = seq.long()
        if torch.is_grad_enabled() and (seq.requires_grad or any(p.requires_grad for p in self.parameters())):
            from ..autograd import backbone_with_grad
            return backbone_with_grad(self, seq.contiguous(), t.to(torch.float32), getattr(self, 'train_dtype', 'fp32'))
        tbias = self.time_bias(t.to(torch.float32))
        if t.shape[0] not in (1, seq.shape[0]):
            raise ValueError(f't: one value per batch row or one shared value (got {t.shape[0]} for batch {seq.shape[0]})')
        return self.forward_f32(seq.contiguous(), tbias, tbias.stride(0) if t.shape[0] == seq.shape[0] and t.shape[0] > 1 else 0)
