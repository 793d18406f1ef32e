"""torch.autograd glue over the HIP forward/backward kernels.

The reference differentiates CNNModel.forward, _subs_parameterization and the Gumbel-softmax
reverse step with torch autograd (diffusion_gosai_update.py:552-630; fine_tune,
finetune_reward_bp.py:60-93).  Here every differentiable block is ONE autograd.Function whose
forward and backward are chains of calls into libdrakes_hip.so; torch only owns the buffers and
the graph.  What is saved per denoiser forward: the 21 residual-stream snapshots, the LayerNorm
statistics, the uint8 ReLU masks and final_conv.0's output (the LayerNorm output h is recomputed
in the backward pass).
"""
import torch

from . import ops

C, V = ops.C, ops.V


def _zeros(*shape, like):
    return torch.zeros(*shape, dtype=torch.float32, device=like.device)


class _ZeroPool:
    """ONE zero-filled fp32 buffer per backward call, carved into the ~90 gradient accumulators the kernels add into
    (one fill launch instead of one per accumulator; pieces start on 16-byte boundaries)."""

    def __init__(self, numel, like):
        self.buf = torch.zeros(numel, dtype=torch.float32, device=like.device)
        self.off = 0

    def __call__(self, *shape):
        n = 1
        for d in shape:
            n *= d
        if self.off + n > self.buf.numel():          # never on the sized paths; keeps the pool safe under changes
            return torch.zeros(*shape, dtype=torch.float32, device=self.buf.device)
        v = self.buf[self.off:self.off + n].view(*shape)
        self.off += (n + 3) // 4 * 4
        return v


def flat_params(bb):
    """All parameters of the denoiser as ONE differentiable fp32 vector (torch.cat of the flattened parameters).

    A caller that runs many differentiable forwards between two optimizer steps (the K Gumbel-softmax steps of
    _sample_finetune_gradient) builds it once and hands it to backbone_with_grad(..., flat=...): every backward call
    then returns ONE gradient vector, the engine adds K vectors instead of K x 129 tensors (one elementwise launch per
    parameter and call: 6 300 launches, ~11 ms per pass at K = 50), and the cat's backward hands each parameter its
    slice once.  The forward never reads the vector's values (it uses the packed kernel-layout copies): it only carries
    the graph."""
    return torch.cat([p.reshape(-1) for p in bb.parameters()])


def _flat_grad(ctx, cls, g, pool, want_w, nb):
    """The gradients in `g` (views of the backward call's zero pool) gathered into parameter order: one index_select.
    The pool's layout is the same on every call of one configuration, so the index vector is built once."""
    bb = ctx.bb
    cache = bb.__dict__.setdefault('_flat_idx', {})
    key = (cls, bool(want_w), int(nb), pool.buf.numel(), tuple(p.requires_grad for p in bb.parameters()))
    idx = cache.get(key)
    if idx is None:
        base, n = pool.buf.data_ptr(), pool.buf.numel()
        addr = torch.arange(n, dtype=torch.int64)
        parts = []
        for name, prm in bb.named_parameters():
            t = g.get(name) if prm.requires_grad else None
            if t is None:
                parts.append(torch.zeros(prm.numel(), dtype=torch.int64))      # element 0 of the pool: never written
                continue
            off = (t.data_ptr() - base) // 4
            inside = 0 <= off and t.numel() == prm.numel() and tuple(t.shape) == tuple(prm.shape) and \
                off + sum((d - 1) * st for d, st in zip(t.shape, t.stride())) < n
            if not inside:                                                     # (pool overflow, a narrowed copy): no gather
                parts = None
                break
            parts.append(addr.as_strided(tuple(t.shape), tuple(t.stride()), off).reshape(-1))
        idx = torch.cat(parts).to(pool.buf.device) if parts is not None else False
        cache[key] = idx
    if idx is False:
        return torch.cat([(g[k] if (k in g and g[k] is not None and prm.requires_grad) else
                           torch.zeros_like(prm)).reshape(-1) for k, prm in bb.named_parameters()])
    return pool.buf.index_select(0, idx)


class _BackboneFn(torch.autograd.Function):
    """logits = CNNModel(seq, sigma).  sigma has one element per batch row, or ONE element shared by
    the whole batch (the sampler's case: all rows sit at the same time, :527-528)."""

    @staticmethod
    def forward(ctx, bb, seq, sigma, mode, *params):
        flat_mode, use_x3 = mode
        ctx.flat_mode = bool(flat_mode)
        pk = bb.packed()
        # 'bf16x3': the convolutions (forward and input gradient) at fp32 grade on the bf16 matrix cores
        # (drk_conv_fwd_x3); weight gradients and everything else stay on the exact-fp32 kernels
        x3 = bb.packed_x3() if use_x3 else None
        conv = (lambda hh, i, *a, **k: ops.conv_fwd_x3(hh, x3['conv_w3'][i], *a, **k)) if x3 else \
               (lambda hh, i, *a, **k: ops.conv_fwd_f32(hh, pk['conv_w'][i], *a, **k))
        seq = seq.contiguous()
        B, L = seq.shape[0], seq.shape[1]
        n = bb.num_layers
        sigma = sigma.to(torch.float32).contiguous()
        nb = sigma.shape[0]
        assert nb in (1, B), 'sigma: one value per row or one shared value'
        temb = ops.time_embed_fwd(sigma, pk['four_W'], pk['temb_w'], pk['temb_b'])
        tbias = ops.time_bias_fwd(temb, pk['dense_w'], pk['dense_b'])
        tb_stride = tbias.stride(0) if nb == B and B > 1 else 0
        feats = [ops.stem_fwd(seq, pk['stem_w'], pk['stem_b'])]
        means, rstds, masks = [], [], []
        h = torch.empty(B, L, C, dtype=torch.float32, device=seq.device)
        for i in range(n):
            mean = torch.empty(B * L, dtype=torch.float32, device=seq.device)
            rstd = torch.empty_like(mean)
            ops.ln_fwd(feats[i], tbias[0, i], tb_stride, pk['ln_g'][i], pk['ln_b'][i], out=h, mean=mean, rstd=rstd, width=bb.width)
            mask = torch.empty(B, L, C, dtype=torch.uint8, device=seq.device)
            feats.append(conv(h, i, pk['conv_b'][i], feats[i], bb.dilation(i), relu=True, relu_mask=mask))
            means.append(mean), rstds.append(rstd), masks.append(mask)
        if x3:
            y = ops.conv_fwd_x3(feats[n], x3['head0_w3'], pk['head0_b'], None, 0, relu=True)
        else:
            y = ops.conv_fwd_f32(feats[n], pk['head0_w'], pk['head0_b'], None, 0, relu=True)
        logits = ops.head_out_fwd(y, pk['head2_w'], pk['head2_b'])
        ctx.x3 = x3
        ctx.bb, ctx.pk, ctx.seq, ctx.sigma, ctx.temb, ctx.tbias, ctx.tb_stride = bb, pk, seq, sigma, temb, tbias, tb_stride
        ctx.feats, ctx.means, ctx.rstds, ctx.masks, ctx.y = feats, means, rstds, masks, y
        ctx.names = [k for k, _ in bb.named_parameters()]
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        bb, pk, seq = ctx.bb, ctx.pk, ctx.seq
        feats, means, rstds, masks, y = ctx.feats, ctx.means, ctx.rstds, ctx.masks, ctx.y
        B, L = seq.shape[0], seq.shape[1]
        M, n = B * L, bb.num_layers
        nb = ctx.sigma.shape[0]
        dlogits = dlogits.contiguous()
        g = {}
        want_w = any(ctx.needs_input_grad[4:])     # a frozen model (the KL reference model) only passes the input gradient on
        zeros = _ZeroPool((n * 9 * C * C if want_w else 0) + n * nb * C + (n + 2) * C * C + 16 * V * C + (6 * n + 16 + nb) * C + 4, dlogits)
        zeros(4)                                   # element 0: the all-zero slot of _flat_grad
        # head
        dy = torch.empty_like(y)
        dw2, db2 = zeros(V, C), zeros(V)
        ops.head_out_bwd(dlogits, y, pk['head2_w'], dy, dw2, db2)
        dpre = ops.relu_bwd(dy, y, out=dy)
        dw0 = zeros(1, C, C)
        db0 = zeros(1, C)
        if want_w:
            ops.conv_wgrad_f32(feats[n], dpre, dw0, 0, db0[0])
        x3 = ctx.x3
        if x3:
            gr = ops.conv_fwd_x3(dpre, x3['head0_w3T'], None, None, 0, relu=False)
        else:
            gr = ops.conv_fwd_f32(dpre, pk['head0_wT'], None, None, 0, relu=False)
        g['final_conv.2.weight'], g['final_conv.2.bias'] = dw2.unsqueeze(-1), db2
        g['final_conv.0.weight'], g['final_conv.0.bias'] = dw0.permute(2, 1, 0), db0[0]
        # layers
        dtb = zeros(n, nb, C)
        h = torch.empty_like(y)
        dc = torch.empty_like(y)
        dh = torch.empty_like(y)
        dxin = torch.empty_like(y)
        for i in reversed(range(n)):
            d = bb.dilation(i)
            ops.mask_mul(gr, masks[i], out=dc)
            if want_w:
                ops.ln_fwd(feats[i], ctx.tbias[0, i], ctx.tb_stride, pk['ln_g'][i], pk['ln_b'][i], out=h, width=bb.width)
                dwt, dbias = zeros(9, C, C), zeros(C)
                ops.conv_wgrad_f32(h, dc, dwt, d, dbias)          # the bias gradient rides on the centre tap
                g[f'convs.{i}.weight'], g[f'convs.{i}.bias'] = dwt.permute(2, 1, 0), dbias
            if x3:
                ops.conv_fwd_x3(dc, x3['conv_w3T'][i], None, None, d, relu=False, out=dh)
            else:
                ops.conv_fwd_f32(dc, pk['conv_wT'][i], None, None, d, relu=False, out=dh)
            dgam, dbet = zeros(C), zeros(C)
            gnew = torch.empty_like(y)
            shared_tb = want_w and not ctx.tb_stride          # one time-bias row for the batch: its gradient is fused in
            ops.ln_bwd(dh, feats[i], ctx.tbias[0, i], ctx.tb_stride, pk['ln_g'][i], means[i], rstds[i], gr, dxin, gnew,
                       dgam, dbet, dtb[i, 0] if shared_tb else None, width=bb.width)
            g[f'norms.{i}.weight'], g[f'norms.{i}.bias'] = dgam, dbet
            if want_w and ctx.tb_stride:
                ops.colsum128(dxin, B, L, dtb[i])
            gr = gnew
        # stem
        dpre = ops.relu_bwd(gr, feats[0], out=gr)
        if want_w:
            g['linear.bias'] = ops.colsum128(dpre, 1, M, zeros(1, C))[0]
        dw_t = zeros(9, V, C)
        dseq = None
        if seq.dtype != torch.int64:
            dseq = torch.empty_like(seq)
        ops.stem_bwd(dpre, seq, pk['stem_w'], dw_t, dseq)
        g['linear.weight'] = dw_t.permute(2, 1, 0)
        # time path
        d_dense_w, d_dense_b = zeros(n, C, C), zeros(n, C)
        d_lin_w, d_lin_b = zeros(C, C), zeros(C)
        if want_w:
            ops.time_bwd(dtb.permute(1, 0, 2).contiguous(), ctx.sigma, pk['four_W'], ctx.temb, pk['dense_w'], d_dense_w,
                         d_dense_b, d_lin_w, d_lin_b, zeros(nb, C))
        g['time_embedder.1.weight'], g['time_embedder.1.bias'] = d_lin_w, d_lin_b
        for i in range(n):
            g[f'time_layers.{i}.dense.weight'], g[f'time_layers.{i}.dense.bias'] = d_dense_w[i], d_dense_b[i]
        # (a model narrower than the kernels' 128 lanes: its gradients are the leading slices of the padded ones)
        ctx.feats = ctx.masks = ctx.means = ctx.rstds = ctx.y = None
        if ctx.flat_mode:
            g = {k: bb.narrow(k, v) for k, v in g.items()}
            return (None, dseq if ctx.needs_input_grad[1] else None, None, None,
                    _flat_grad(ctx, 'f32', g, zeros, want_w, nb) if want_w else None)
        grads = [bb.narrow(k, g.get(k)) if ctx.needs_input_grad[4 + j] else None for j, k in enumerate(ctx.names)]
        return (None, dseq if ctx.needs_input_grad[1] else None, None, None, *grads)


class GradSink:
    """Where the bf16 backward leaves everything the PARAMETER gradients need, so that they are formed once per chain of
    forwards instead of once per forward (the optimizer reads them only after the whole backward pass,
    finetune_reward_bp.py:88-93):
      * per recorded forward, the bf16 operand pairs of every convolution's weight gradient — hs[i] (the LayerNorm
        outputs the forward saved) and dcs[i] (the gated output gradients the backward produced) — by reference;
      * per distinct sigma, one fp32 accumulator row [W] for the column-sum quantities (LayerNorm affine, conv biases, time
        biases, the 128 -> 5 head), filled by ONE ordered slab reduction per backward call;
      * stem weight / bias gradients, accumulated by the stem kernels' atomics.
    flush() then runs ONE drk_conv_wgrad_bf16_multi per layer over all recorded forwards (K x B x L rows), one
    drk_time_bwd per distinct sigma, and returns {parameter name: gradient}.  Deterministic for the conv weights (99 % of
    the parameters)."""

    def __init__(self, bb):
        self.bb = bb
        n = bb.num_layers
        self.n = n
        self.head_off = (n + 1) * 4 * C                  # layer slots 0..n-1, slot n = the head's input gradient launch
        self.W = self.head_off + int(_lib_slab())        # + drk_head_bwd_bf16's slab (= drk_denoiser_bwd_bf16_slab_floats(n))
        self.records = []                                # (B, L, hs block [n+1,B,L,128], dcs block [n+1,B,L,128])
        self.time = {}                                   # sigma key -> (sigma, temb, tbias): the time path runs once per sigma
        self.groups = {}                                 # sigma key -> (sigma, temb, acc [W])
        self.stem_dw = None
        self.stem_db = None
        self.stem_records = []                           # (B, L, dpre [B,L,128], the call's input): the stem's gradients wait too
        self.deferred_stem = False                       # set by grad_chain: a sink that outlives one backward call

    def group(self, key, sigma, temb):
        """The accumulator row of the forwards that ran at this sigma (key: its value as a Python number, known to the
        caller without a device read; None: this forward's own row)."""
        key = id(sigma) if key is None else key
        g = self.groups.get(key)
        if g is None:
            g = (sigma, temb, torch.zeros(self.W, dtype=torch.float32, device=temb.device))
            self.groups[key] = g
        return g[2]

    def stem(self, dev):
        if self.stem_dw is None:
            self.stem_dw = torch.zeros(9, V, C, dtype=torch.float32, device=dev)
            self.stem_db = torch.zeros(1, C, dtype=torch.float32, device=dev)
        return self.stem_dw, self.stem_db

    def flush(self):
        """-> {name: gradient} for every trainable parameter the recorded backward calls reached; clears the records."""
        bb, n = self.bb, self.n
        g = {}
        if not self.groups:
            return g
        pk = bb.packed()
        dev = next(iter(self.groups.values()))[2].device
        # ---- convolution weights: one launch per layer and shape over all recorded forwards
        dwt = torch.zeros(9 * n + 1, C, C, dtype=torch.float32, device=dev)
        by_shape = {}
        for B, L, hs, dcs in self.records:
            by_shape.setdefault((B, L), []).append((hs, dcs))
        for (B, L), recs in by_shape.items():
            K = len(recs)
            step = B * L * C * 2                       # bytes between the layers of a block
            ptrs = torch.tensor([[[r[w].data_ptr() + i * step for r in recs] for i in range(n + 1)] for w in (0, 1)],
                                dtype=torch.int64, pin_memory=True).to(dev, non_blocking=True)
            for i in range(n):
                ops.conv_wgrad_bf16_multi(ptrs[0, i], ptrs[1, i], K, dwt[9 * i:9 * i + 9], B, L, bb.dilation(i))
            ops.conv_wgrad_bf16_multi(ptrs[0, n], ptrs[1, n], K, dwt[9 * n:], B, L, 0)
        for i in range(n):
            g[f'convs.{i}.weight'] = dwt[9 * i:9 * i + 9].permute(2, 1, 0)
        g['final_conv.0.weight'] = dwt[9 * n:].permute(2, 1, 0)
        # ---- column-sum quantities
        accs = [a for _, _, a in self.groups.values()]
        tot = accs[0] if len(accs) == 1 else torch.stack(accs).sum(0)
        lay = tot[:self.head_off].view(n + 1, 4, C)
        for i in range(n):
            g[f'norms.{i}.weight'], g[f'norms.{i}.bias'] = lay[i, 0], lay[i, 1]
            g[f'convs.{i}.bias'] = lay[i + 1, 3]            # the gate sums of the launch ABOVE layer i
        hd = tot[self.head_off:]
        g['final_conv.2.weight'] = hd[:V * C].view(V, C).unsqueeze(-1)
        g['final_conv.0.bias'] = hd[V * C:(V + 1) * C]
        g['final_conv.2.bias'] = hd[(V + 1) * C:(V + 1) * C + V]
        # ---- stem: the recorded calls of one shape and input kind in one launch
        by_kind = {}
        for B, L, dpre, seq in self.stem_records:
            by_kind.setdefault((B, L, seq.dtype == torch.int64), []).append((dpre, seq))
        for (B, L, tok), recs in by_kind.items():
            dw_t, db = self.stem(dev)
            ptrs = torch.tensor([[r[w].data_ptr() for r in recs] for w in (0, 1)], dtype=torch.int64,
                                pin_memory=True).to(dev, non_blocking=True)
            ops.stem_wgrad_multi(ptrs[0], ptrs[1], tok, len(recs), dw_t, db, B, L)
        if self.stem_dw is not None:
            g['linear.weight'] = self.stem_dw.permute(2, 1, 0)
            g['linear.bias'] = self.stem_db[0]
        # ---- time path: one backward per distinct sigma (the time-bias gradients of its forwards, summed)
        d_dense_w = torch.zeros(n, C, C, dtype=torch.float32, device=dev)
        d_dense_b = torch.zeros(n, C, dtype=torch.float32, device=dev)
        d_lin_w = torch.zeros(C, C, dtype=torch.float32, device=dev)
        d_lin_b = torch.zeros(C, dtype=torch.float32, device=dev)
        for sigma, temb, acc in self.groups.values():
            dtb = acc[:n * 4 * C].view(n, 4, C)[:, 2].contiguous().view(1, n, C)
            ops.time_bwd(dtb, sigma, pk['four_W'], temb, pk['dense_w'], d_dense_w, d_dense_b, d_lin_w, d_lin_b,
                         torch.zeros(1, C, dtype=torch.float32, device=dev))
        g['time_embedder.1.weight'], g['time_embedder.1.bias'] = d_lin_w, d_lin_b
        for i in range(n):
            g[f'time_layers.{i}.dense.weight'], g[f'time_layers.{i}.dense.bias'] = d_dense_w[i], d_dense_b[i]
        self.records, self.groups, self.time, self.stem_dw, self.stem_db = [], {}, {}, None, None
        self.stem_records = []
        return g


def _lib_slab():
    from . import _lib
    return _lib.load().drk_head_bwd_bf16_slab_floats()


class _ChainFn(torch.autograd.Function):
    """The graph node through which a GradSink's gradients reach the parameters: its output (one element, never read)
    is handed to every forward of the chain in place of the parameters, so autograd runs this backward once, after the
    last of them."""

    @staticmethod
    def forward(ctx, sink, *params):
        ctx.sink = sink
        ctx.names = [k for k, _ in sink.bb.named_parameters()]
        return params[0].new_zeros(1)

    @staticmethod
    def backward(ctx, _g):
        g = ctx.sink.flush()
        return (None, *[g.get(k) if ctx.needs_input_grad[1 + j] else None for j, k in enumerate(ctx.names)])


def grad_chain(bb):
    """(sink, anchor) for a chain of differentiable bf16 forwards between two optimizer steps (the K Gumbel-softmax steps
    of _sample_finetune_gradient): pass both to backbone_with_grad(..., dtype='bf16', flat=anchor, sink=sink)."""
    sink = GradSink(bb)
    sink.deferred_stem = True
    return sink, _ChainFn.apply(sink, *bb.parameters())


class _BackboneBf16Fn(torch.autograd.Function):
    """logits = CNNModel(seq, sigma) on the bf16-MFMA layer kernels, differentiable: bf16 LayerNorm outputs and conv
    operands (forward, input gradient, weight gradient), fp32 accumulation, fp32 residual stream and residual-gradient
    stream, fp32 LayerNorm statistics and backward, fp32 stem / 128->5 head / time path.  Saved per forward: the 21
    residual snapshots (fp32), the 21 LayerNorm outputs (bf16: the weight gradient's operand), LayerNorm statistics and
    uint8 ReLU masks.
    Backward (round 3): ONE launch per layer — drk_layer_bwd_bf16 = input gradient + LayerNorm backward + residual add +
    the gate of the layer below — and the parameter gradients deferred to a GradSink (one weight-gradient launch per
    layer over the whole chain).  A per-row time bias (per-row sigma) takes the round-2 launch sequence."""

    @staticmethod
    def forward(ctx, bb, seq, sigma, sink, *params):
        ctx.sink, ctx.sigma_key = (None, None) if sink is None else sink
        sink = ctx.sink
        from . import fastpath
        pk, st = bb.packed(), fastpath._packed(bb)
        seq = seq.contiguous()
        B, L = seq.shape[0], seq.shape[1]
        n = bb.num_layers
        dev = seq.device
        sigma = sigma.to(torch.float32).contiguous()
        nb = sigma.shape[0]
        assert nb in (1, B), 'sigma: one value per row or one shared value'
        hit = sink.time.get(ctx.sigma_key) if (sink is not None and ctx.sigma_key is not None) else None
        if hit is not None:                         # the time path of this sigma ran earlier in the chain (same weights)
            sigma, temb, tbias = hit
        else:
            temb = ops.time_embed_fwd(sigma, pk['four_W'], pk['temb_w'], pk['temb_b'])
            tbias = ops.time_bias_fwd(temb, pk['dense_w'], pk['dense_b'])
            if sink is not None and ctx.sigma_key is not None:
                sink.time[ctx.sigma_key] = (sigma, temb, tbias)
        tb_stride = tbias.stride(0) if nb == B and B > 1 else 0
        ctx.bb, ctx.pk, ctx.seq, ctx.sigma, ctx.temb, ctx.tbias, ctx.tb_stride = bb, pk, seq, sigma, temb, tbias, tb_stride
        ctx.names = [k for k, _ in bb.named_parameters()]
        if tb_stride:
            assert sink is None, 'a gradient chain shares one sigma per forward'
            return _BackboneBf16Fn._forward_rows(ctx, st)
        M = B * L
        ctx.per_layer = bool(ops.DENOISER_PER_LAYER)        # the backward runs in the mode that filled the saved blocks
        feats, masks = ops.denoiser_saved_buffers(B, L, n, dev, ctx.per_layer)
        hs = torch.empty(n + 1, B, L, C, dtype=torch.bfloat16, device=dev)
        stats = torch.empty(2, n, M, dtype=torch.float32, device=dev)
        y = torch.empty(B, L, C, dtype=torch.float32, device=dev)
        logits = torch.empty(B, L, V, dtype=torch.float32, device=dev)
        ops.denoiser_fwd_bf16(seq, pk, st, tbias[0], feats, hs, masks, stats[0], stats[1], y, logits, per_layer=ctx.per_layer)
        ctx.st = st
        ctx.feats, ctx.hs, ctx.means, ctx.rstds, ctx.masks, ctx.y = feats, hs, stats[0], stats[1], masks, y
        if _BackboneBf16Fn.gate_log is not None:   # tests: the ReLU gates this forward took (see oracle/bf16_emulation.py)
            fused = ops.denoiser_is_fused(L, n, ctx.per_layer)
            _BackboneBf16Fn.gate_log.append((fastpath.saved_masks_rows(masks, B, L, n, fused), y,
                                             fastpath.saved_feats_rows(feats, B, L, n, fused)[0]))
        return logits

    @staticmethod
    def _forward_rows(ctx, st):
        """A time bias per batch row (per-row sigma): the per-launch sequence of round 2."""
        bb, pk, seq, tbias, tb_stride = ctx.bb, ctx.pk, ctx.seq, ctx.tbias, ctx.tb_stride
        vec, wstream = st['vecs'], st['wstream']
        B, L = seq.shape[0], seq.shape[1]
        n = bb.num_layers
        dev = seq.device
        f32 = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        b16 = lambda: torch.empty(B, L, C, dtype=torch.bfloat16, device=dev)
        feats = [ops.stem_fwd(seq, pk['stem_w'], pk['stem_b'])]
        means, rstds, masks = [f32(B * L)], [f32(B * L)], []
        hs = [ops.ln_fwd_bf16(feats[0], tbias[0, 0], tb_stride, vec[1], vec[2], mean=means[0], rstd=rstds[0])]
        for i in range(n):
            last = i + 1 == n
            mask = torch.empty(B, L, C, dtype=torch.uint8, device=dev)
            if not last:
                means.append(f32(B * L)), rstds.append(f32(B * L))
            hs.append(b16())
            feats.append(ops.layer_bf16(
                hs[i], wstream[9 * i:], f32(B, L, C), feat=feats[i], h_out=hs[i + 1], bias=vec[3 * (i + 1)],
                tbias=None if last else tbias[0, i + 1], tb_stride=tb_stride, gamma=None if last else vec[3 * (i + 1) + 1],
                beta=None if last else vec[3 * (i + 1) + 2], taps=9, dilation=bb.dilation(i), relu=True, relu_mask=mask,
                mean=None if last else means[i + 1], rstd=None if last else rstds[i + 1]))
            masks.append(mask)
        # hs[n] = bf16(feat_n): final_conv.0 (1x1) + ReLU on the same kernel, then the exact-fp32 128 -> 5 conv
        y = ops.layer_bf16(hs[n], wstream[9 * n:], f32(B, L, C), bias=vec[3 * n + 1], taps=1, dilation=0, relu=True)
        logits = ops.head_out_fwd(y, pk['head2_w'], pk['head2_b'])
        ctx.feats, ctx.hs, ctx.means, ctx.rstds, ctx.masks, ctx.y = feats, hs, means, rstds, masks, y
        if _BackboneBf16Fn.gate_log is not None:
            _BackboneBf16Fn.gate_log.append((masks, y, feats[0]))
        return logits

    gate_log = None          # tests set this to a list; every bf16 forward under autograd then appends (masks, y, feats[0])

    @staticmethod
    def backward(ctx, dlogits):
        if ctx.tb_stride:
            return _BackboneBf16Fn._backward_unfused(ctx, dlogits)
        from . import fastpath
        bb, pk, st, seq = ctx.bb, ctx.pk, ctx.st, ctx.seq
        B, L = seq.shape[0], seq.shape[1]
        M, n = B * L, bb.num_layers
        dev = dlogits.device
        dlogits = dlogits.contiguous()
        want_w = any(ctx.needs_input_grad[4:])     # a frozen model (the KL reference model) only passes the input gradient on
        sink = ctx.sink if ctx.sink is not None else (GradSink(bb) if want_w else None)
        dcs = torch.empty(n + 1 if want_w else 2, B, L, C, dtype=torch.bfloat16, device=dev)
        gr = torch.empty(B, L, C, dtype=torch.float32, device=dev)      # the residual-gradient stream, updated in place
        dseq = torch.empty_like(seq) if (seq.dtype != torch.int64 and ctx.needs_input_grad[1]) else None
        kw = {}
        if want_w:
            P = ops.workspace(ops.denoiser_bwd_slabs(B, L) * sink.W * 4, dev)   # [slab][W]: every slot written by one launch
            kw = dict(partials=P, acc=sink.group(ctx.sigma_key, ctx.sigma, ctx.temb))
            if not sink.deferred_stem:
                kw['stem_dw'], kw['stem_db'] = sink.stem(dev)
        ops.denoiser_bwd_bf16(dlogits, seq, pk, st, fastpath._packed_T(bb), ctx.tbias[0], ctx.feats, ctx.masks, ctx.means,
                              ctx.rstds, ctx.y, dcs, gr, dx=dseq, per_layer=ctx.per_layer, **kw)
        if want_w:
            sink.records.append((B, L, ctx.hs, dcs))
            if sink.deferred_stem:
                sink.stem_records.append((B, L, gr, seq))       # gr = dL/d(stem pre-activation) now
        ctx.feats = ctx.hs = ctx.masks = ctx.means = ctx.rstds = ctx.y = None
        if ctx.sink is not None or not want_w:        # the chain's _ChainFn hands the parameters their gradients
            return (None, dseq, None, None) + (None,) * len(ctx.needs_input_grad[4:])
        g = sink.flush()
        return (None, dseq, None, None, *[g.get(k) if ctx.needs_input_grad[4 + j] else None for j, k in enumerate(ctx.names)])

    @staticmethod
    def _backward_unfused(ctx, dlogits):
        from . import fastpath
        bb, pk, seq = ctx.bb, ctx.pk, ctx.seq
        feats, hs, means, rstds, masks, y = ctx.feats, ctx.hs, ctx.means, ctx.rstds, ctx.masks, ctx.y
        B, L = seq.shape[0], seq.shape[1]
        M, n = B * L, bb.num_layers
        nb = ctx.sigma.shape[0]
        dlogits = dlogits.contiguous()
        g = {}
        want_w = any(ctx.needs_input_grad[4:])
        wT = fastpath._packed_T(bb)
        zeros = _ZeroPool((n * 9 * C * C if want_w else 0) + n * nb * C + (n + 2) * C * C + 16 * V * C + (6 * n + 16 + nb) * C + 4, dlogits)
        zeros(4)                                   # element 0: the all-zero slot of _flat_grad
        # head: exact-fp32 128 -> 5 backward, then final_conv.0 on the bf16 kernels
        dy = torch.empty_like(y)
        dw2, db2 = zeros(V, C), zeros(V)
        ops.head_out_bwd(dlogits, y, pk['head2_w'], dy, dw2, db2)
        dw0, db0 = zeros(1, C, C), zeros(C)
        dc = ops.gate_bf16(dy, y=y, colsum=db0 if want_w else None)
        if want_w:
            ops.conv_wgrad_bf16(hs[n], dc, dw0, 0)
        gr = ops.layer_bf16(dc, wT[9 * n:], dy, taps=1, dilation=0, relu=False)       # dy's storage becomes dL/dfeat_n
        g['final_conv.2.weight'], g['final_conv.2.bias'] = dw2.unsqueeze(-1), db2
        g['final_conv.0.weight'], g['final_conv.0.bias'] = dw0.permute(2, 1, 0), db0
        dtb = zeros(n, nb, C)
        dh = torch.empty_like(y)
        dxin = torch.empty_like(y) if (want_w and ctx.tb_stride) else None      # only the per-row time-bias gradient reads it
        dbias = zeros(C)
        ops.gate_bf16(gr, relu_mask=masks[n - 1], colsum=dbias if want_w else None, out=dc)
        dc_next = torch.empty_like(dc)
        for i in reversed(range(n)):
            d = bb.dilation(i)
            if want_w:
                dwt = zeros(9, C, C)
                ops.conv_wgrad_bf16(hs[i], dc, dwt, d)
                g[f'convs.{i}.weight'], g[f'convs.{i}.bias'] = dwt.permute(2, 1, 0), dbias
            ops.layer_bf16(dc, wT[9 * i:], dh, taps=9, dilation=d, relu=False)
            dgam, dbet = zeros(C), zeros(C)
            gnew = torch.empty_like(y)
            shared_tb = want_w and not ctx.tb_stride
            dbias = zeros(C)                                        # of layer i - 1, filled by the fused hand-over
            ops.ln_bwd(dh, feats[i], ctx.tbias[0, i], ctx.tb_stride, pk['ln_g'][i], means[i], rstds[i], gr, dxin, gnew,
                       dgam, dbet, dtb[i, 0] if shared_tb else None, next_mask=masks[i - 1] if i else None,
                       dc_next=dc_next if i else None, dbias_next=dbias if (i and want_w) else None)
            g[f'norms.{i}.weight'], g[f'norms.{i}.bias'] = dgam, dbet
            if want_w and ctx.tb_stride:
                ops.colsum128(dxin, B, L, dtb[i])
            gr = gnew
            dc, dc_next = dc_next, dc
        # stem and time path: the fp32 kernels
        dpre = ops.relu_bwd(gr, feats[0], out=gr)
        if want_w:
            g['linear.bias'] = ops.colsum128(dpre, 1, M, zeros(1, C))[0]
        dw_t = zeros(9, V, C)
        dseq = None
        if seq.dtype != torch.int64:
            dseq = torch.empty_like(seq)
        ops.stem_bwd(dpre, seq, pk['stem_w'], dw_t, dseq)
        g['linear.weight'] = dw_t.permute(2, 1, 0)
        d_dense_w, d_dense_b = zeros(n, C, C), zeros(n, C)
        d_lin_w, d_lin_b = zeros(C, C), zeros(C)
        if want_w:
            ops.time_bwd(dtb.permute(1, 0, 2).contiguous(), ctx.sigma, pk['four_W'], ctx.temb, pk['dense_w'], d_dense_w,
                         d_dense_b, d_lin_w, d_lin_b, zeros(nb, C))
        g['time_embedder.1.weight'], g['time_embedder.1.bias'] = d_lin_w, d_lin_b
        for i in range(n):
            g[f'time_layers.{i}.dense.weight'], g[f'time_layers.{i}.dense.bias'] = d_dense_w[i], d_dense_b[i]
        ctx.feats = ctx.hs = ctx.masks = ctx.means = ctx.rstds = ctx.y = None
        grads = [g.get(k) if ctx.needs_input_grad[4 + j] else None for j, k in enumerate(ctx.names)]
        return (None, dseq if ctx.needs_input_grad[1] else None, None, None, *grads)


def backbone_with_grad(bb, seq, sigma, dtype='fp32', flat=None, sink=None, sigma_key=None):
    """dtype 'bf16': the bf16-MFMA layer kernels (gradients within the bf16 bar); 'fp32': the exact-fp32 kernels;
    'bf16x3': those with the convolutions (forward and input gradient) at fp32 grade on the bf16 matrix cores.
    flat: flat_params(bb), built once by a caller that runs many forwards per backward pass (see there); on the bf16 path
    (sink, flat) = grad_chain(bb) instead, and sigma_key = sigma's value as a Python number when the caller knows it."""
    if dtype is True:
        dtype = 'bf16'
    if dtype == 'bf16':      # sink, flat = grad_chain(bb): parameter gradients deferred to the end of the chain
        if sink is not None:
            return _BackboneBf16Fn.apply(bb, seq, sigma, (sink, sigma_key), flat)
        return _BackboneBf16Fn.apply(bb, seq, sigma, None, *bb.parameters())
    x3 = dtype == 'bf16x3'
    if flat is not None:
        return _BackboneFn.apply(bb, seq, sigma, (True, x3), flat)
    return _BackboneFn.apply(bb, seq, sigma, (False, x3), *bb.parameters())


class _SubsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, xt):
        ctx.save_for_backward(logits, xt)
        return ops.subs_fwd(logits.contiguous(), xt.contiguous())

    @staticmethod
    def backward(ctx, dlogp):
        logits, xt = ctx.saved_tensors
        return ops.subs_bwd(logits.contiguous(), xt.contiguous(), dlogp.contiguous()), None


def subs_with_grad(logits, xt):
    return _SubsFn.apply(logits, xt)


class _SoftStepFn(torch.autograd.Function):
    """(x_next, soft_copy_flag) = soft reverse step(logits, x_soft); noise is recomputed in the backward."""

    @staticmethod
    def forward(ctx, logits, x_soft, mc_t, mc_s, temp, cft, seed, step, sample_offset, uniforms, mc_rows=None):
        logits, x_soft = logits.contiguous(), x_soft.contiguous()
        x_next, flag = ops.reverse_step_soft_fwd(logits, x_soft, mc_t, mc_s, temp, cft, seed, step, sample_offset, uniforms,
                                                 mc_rows)
        ctx.save_for_backward(logits, x_soft)
        ctx.consts = (mc_t, mc_s, temp, cft, seed, step, sample_offset, uniforms, mc_rows)
        return x_next, flag.unsqueeze(-1)

    @staticmethod
    def backward(ctx, g_next, g_flag):
        logits, x_soft = ctx.saved_tensors
        mc_t, mc_s, temp, cft, seed, step, sample_offset, uniforms, mc_rows = ctx.consts
        gf = g_flag.reshape(g_flag.shape[0], g_flag.shape[1]).contiguous() if g_flag is not None else None
        dlogits, dx = ops.reverse_step_soft_bwd(logits, x_soft, g_next.contiguous(), gf, mc_t, mc_s, temp, cft, seed, step,
                                                sample_offset, uniforms, mc_rows)
        return dlogits, dx, None, None, None, None, None, None, None, None, None


def soft_step_with_grad(logits, x_soft, mc_t, mc_s, temp, cft, seed, step, sample_offset, uniforms, mc_rows=None):
    return _SoftStepFn.apply(logits, x_soft, mc_t, mc_s, temp, cft, seed, step, sample_offset, uniforms, mc_rows)


class _KLStepFn(torch.autograd.Function):
    """kl[b] of one reverse step (finetune_reward_bp.py:67-74).  Differentiable w.r.t. logp and, on soft steps,
    last_x, copy_flag (both carry a graph there, diffusion_gosai_update.py:580-586) and logp_old (a function of last_x
    through the frozen model, finetune_reward_bp.py:68)."""

    @staticmethod
    def forward(ctx, logp, logp_old, last_x, flag, mc_t):
        logp, logp_old, last_x, flag = logp.contiguous(), logp_old.contiguous(), last_x.contiguous(), flag.contiguous()
        ctx.save_for_backward(logp, logp_old, last_x, flag)
        ctx.mc_t = mc_t
        ctx.flag_shape = flag.shape
        return ops.kl_step_fwd(logp, logp_old, last_x, flag, mc_t)

    @staticmethod
    def backward(ctx, g):
        logp, logp_old, last_x, flag = ctx.saved_tensors
        want_old, want_dx = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        want_df = ctx.needs_input_grad[3] and flag.dtype == torch.float32
        dlogp, dold, dx, dflag = ops.kl_step_bwd(logp, logp_old, last_x, flag, g.contiguous(), ctx.mc_t, want_old, want_dx,
                                                 want_df)
        return dlogp, dold, dx, (dflag.reshape(ctx.flag_shape) if dflag is not None else None), None


def kl_step_with_grad(logp, logp_old, last_x, flag, mc_t):
    return _KLStepFn.apply(logp, logp_old, last_x, flag, float(mc_t))


class _SoftStepWideFn(torch.autograd.Function):
    """(x_next, flag) of the protein sampler's Gumbel-softmax step (fm_utils.py:159-192); noise recomputed in backward."""

    @staticmethod
    def forward(ctx, logits, x_soft, d_t, mc_s, temp, gtemp, mask_index, seed, step, sample_offset, uniforms):
        logits, x_soft = logits.contiguous(), x_soft.contiguous()
        x_next, flag = ops.reverse_step_soft_wide_fwd(logits, x_soft, d_t, mc_s, temp, gtemp, mask_index, seed, step,
                                                      sample_offset, uniforms)
        ctx.save_for_backward(logits, x_soft)
        ctx.consts = (d_t, mc_s, temp, gtemp, mask_index, seed, step, sample_offset, uniforms)
        return x_next, flag.unsqueeze(-1)

    @staticmethod
    def backward(ctx, g_next, g_flag):
        logits, x_soft = ctx.saved_tensors
        d_t, mc_s, temp, gtemp, mask_index, seed, step, sample_offset, uniforms = ctx.consts
        gf = g_flag.reshape(g_flag.shape[0], g_flag.shape[1]).contiguous() if g_flag is not None else None
        dlogits, dx = ops.reverse_step_soft_wide_bwd(logits, x_soft, g_next.contiguous(), gf, d_t, mc_s, temp, gtemp,
                                                     mask_index, seed, step, sample_offset, uniforms)
        return (dlogits, dx) + (None,) * 9


def soft_step_wide_with_grad(logits, x_soft, d_t, mc_s, temp, gtemp, mask_index, seed, step, sample_offset, uniforms):
    return _SoftStepWideFn.apply(logits, x_soft, d_t, mc_s, temp, gtemp, mask_index, seed, step, sample_offset, uniforms)
