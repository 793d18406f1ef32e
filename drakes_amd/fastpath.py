"""Host side of the bf16 fast path: packs the denoiser for drk_sample_bf16 (the fused sampling
loop, drakes_amd/csrc/drk_sample_bf16.hip) and launches it.

Weight stream layout (MFMA-fragment order, read straight from L2 into registers):
    tap = [4 k-steps][4 co tiles][2 row tiles][64 lanes][8 bf16] = 32 KB (see _frag_tap);
    order = for each layer its 9 taps, then final_conv.0.
The stem (MFMA A fragments over K = 8*tap + v padded to 96: [3 k-steps][4 co tiles][2 row tiles][64 lanes][8], see
_frag_stem) and final_conv.2 ([8][128] bf16, rows 0..4) are small and stay resident in LDS for the whole launch.
"""
import ctypes

import torch

from . import _lib, ops

MAX_L = 224
SAMPLE_GENERIC = 1        # include/drakes_hip.h: DRK_SAMPLE_GENERIC
launch_flags = 0          # tests set SAMPLE_GENERIC to compare the two builds of the fused kernel


def _frag_tap(m):
    """[128 co][128 ci] float -> bf16 [4 k-steps][4 co tiles][2 row tiles][64 lanes][8]: lane 16g + n of the wave that
    owns co tile ct holds, for k-step ks and row tile s, W[32ct + 16s + n][64(g&1) + 32(g>>1) + 8ks + 0..7] — the A
    operand of v_mfma_f32_16x16x32_bf16 with the kernel's k-slot <-> channel map (drk_sample_bf16.hip, header)."""
    w = m.to(torch.bfloat16).reshape(4, 2, 16, 2, 2, 4, 8)     # [ct][s][n][g&1][g>>1][ks][j]
    return w.permute(5, 0, 1, 4, 3, 2, 6).contiguous().reshape(4, 4, 2, 64, 8)


def _frag_stem(w):
    """linear.weight [128 co][5 v][9 taps] -> bf16 [3 k-steps][4 co tiles][2 row tiles][64 lanes][8]: the stem as a
    GEMM over K = 8*tap + v (v = 5..7 and taps 9..11 zero).  Lane 16g + n holds, for k-step ks and row tile s,
    W[32ct + 16s + n][v = 0..7][tap = 4ks + g]; the kernel builds the matching one-hot B operand in registers."""
    wk = torch.zeros(128, 12, 8, device=w.device)
    wk[:, :9, :5] = w.permute(0, 2, 1)
    wk = wk.to(torch.bfloat16).reshape(4, 2, 16, 3, 4, 8)      # [ct][s][n][ks][g][j]
    return wk.permute(3, 0, 1, 4, 2, 5).contiguous().reshape(3, 4, 2, 64, 8)


def _frag_taps(m):
    """_frag_tap for a batch of matrices [T][128 co][128 ci] in one pass (the packing runs after every optimizer step)."""
    w = m.to(torch.bfloat16).reshape(-1, 4, 2, 16, 2, 2, 4, 8)     # [T][ct][s][n][g&1][g>>1][ks][j]
    return w.permute(0, 6, 1, 2, 5, 4, 3, 7).contiguous().reshape(-1, 4, 4, 2, 64, 8)


def pack_stream(bb):
    """bb: models.dnaconv.CNNModel -> dict of device tensors the fused kernel reads."""
    dev = bb.linear.weight.device
    n = bb.num_layers
    with torch.no_grad():
        w = torch.stack([bb.convs[i].weight for i in range(n)])                 # [n][co][ci][tap]
        mats = torch.cat([w.permute(0, 3, 1, 2).reshape(9 * n, 128, 128), bb.final_conv[0].weight[:, :, 0][None]])
        head2 = torch.zeros(128, 128, device=dev)
        head2[:ops.V] = bb.final_conv[2].weight[:, :, 0]
        # epilogue blocks of 3 vectors: block 0 = stem bias, gamma_0, beta_0; block k+1 = conv bias_k,
        # gamma_{k+1}, beta_{k+1}; last block = conv bias_{n-1}, final_conv.0 bias, final_conv.2 bias (padded)
        hb2 = torch.zeros(128, device=dev)
        hb2[:ops.V] = bb.final_conv[2].bias
        vec_rows = [bb.linear.bias, bb.norms[0].weight, bb.norms[0].bias]
        for k in range(n):
            if k + 1 < n:
                vec_rows += [bb.convs[k].bias, bb.norms[k + 1].weight, bb.norms[k + 1].bias]
            else:
                vec_rows += [bb.convs[k].bias, bb.final_conv[0].bias, hb2]
        return {
            'wstream': _frag_taps(mats),
            'stem_w': _frag_stem(bb.linear.weight.detach()),
            'head2_w': head2[:8].to(torch.bfloat16).contiguous(),
            'vecs': torch.stack([v.detach().float() for v in vec_rows]).contiguous(),
            'dil': (ctypes.c_int * n)(*[bb.dilation(i) for i in range(n)]),
            'n_layers': n,
        }


def pack_stream_T(bb):
    """The fragment stream of the convolutions' INPUT gradients: per layer the 9 taps of weight.transpose(0,1).flip(2)
    (dh[r][ci] = sum_t sum_co dc[r + (t-4)d][co] * w[co][ci][8-t]), then final_conv.0 transposed."""
    n = bb.num_layers
    with torch.no_grad():
        w = torch.stack([bb.convs[i].weight for i in range(n)])                 # [n][co][ci][tap]
        wt = w.permute(0, 3, 2, 1).flip(1)                                      # [n][8 - tap][ci][co]
        mats = torch.cat([wt.reshape(9 * n, 128, 128), bb.final_conv[0].weight[:, :, 0].t()[None]])
        return _frag_taps(mats)


def _packed_T(bb):
    _need_full_width(bb)
    key = bb._weights_key()
    if getattr(bb, '_streamT_key', None) != key:
        bb._streamT = pack_stream_T(bb)
        bb._streamT_key = key
    return bb._streamT


def _need_full_width(bb):
    if getattr(bb, 'width', 128) != 128:
        raise NotImplementedError(f'the bf16 kernels are written for hidden_dim = 128 (got {bb.width}): run a narrower model '
                                  "on the exact-fp32 kernels (hip_dtype = 'fp32', hip_soft_dtype = None)")


def _packed(bb):
    _need_full_width(bb)
    key = bb._weights_key()
    if getattr(bb, '_stream_key', None) != key:
        bb._stream = pack_stream(bb)
        bb._stream_key = key
    return bb._stream


def _launch(bb, tbias, tb_stride, mc, uniforms, x_init, x_out, logits_out, dump_fwd, B, L, num_steps, noise_removal,
            seed, sample_offset, dbg_feat=None, dbg_layer=0, traj=None):
    pk = _packed(bb)
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    args = (P(pk['wstream']), P(pk['stem_w']), P(pk['head2_w']), P(pk['vecs']), P(tbias), tb_stride, P(mc), P(uniforms), P(x_init), P(x_out),
            P(traj), P(logits_out), dump_fwd, P(dbg_feat), dbg_layer, B, L, num_steps, pk['n_layers'], pk['dil'], 1 if noise_removal else 0, seed,
            sample_offset, launch_flags, ops._stream())
    flop = 2.0 * B * L * (45 * 128 + pk['n_layers'] * 9 * 128 * 128 + 128 * 128 + 128 * 5) * (num_steps + (1 if noise_removal else 0))
    ops.PROFILE.bracket('k_sample_bf16', flop, lambda: _lib.call('drk_sample_bf16', *args))


def _uses_fast_build(bb, L):
    """Mirror of the entry point's choice (drk_sample_bf16.hip): the specialised build runs for 192 < L <= 208 when every
    dilation is in {1..4, 16, 64}."""
    dil = [bb.dilation(i) for i in range(bb.num_layers)]
    return (192 < L <= 208 and all(1 <= d <= 4 or d in (16, 64) for d in dil) and not (launch_flags & SAMPLE_GENERIC))


def executed_flop_fraction(bb, L):
    """MFMA FLOPs one forward of the fused kernel EXECUTES / the dense algorithmic FLOPs of the reference's forward
    (what bench.py's roofline numerator counts).  The kernel computes whole 16-position tiles (L=200 -> 208), runs the
    stem as K = 96 instead of 45 and final_conv.2 as 16 output rows instead of 5, and skips (tile, tap) pairs that only
    multiply padding: every pair of a tap whose shift is >= L in both builds; in the specialised build also the pairs
    with all 16 shifted rows outside [0, 208) for dilations 16 and 64 (live_lo / live_hi in drk_sample_bf16.hip)."""
    n = bb.num_layers
    ntiles = (L + 15) // 16
    fast = _uses_fast_build(bb, L)
    tile_taps = 0
    for i in range(n):
        d = bb.dilation(i)
        for k in range(-4, 5):
            off = k * d
            if abs(off) >= L and off != 0:
                continue                                   # whole tap skipped (both builds)
            if fast and d in (16, 64):
                live = sum(1 for T in range(ntiles) if (16 * T + off < 208 if off > 0 else 16 * T + 15 >= -off))
            else:
                live = ntiles
            tile_taps += live
    mfma = 16 * 16 * 32 * 2                                # FLOPs of one v_mfma_f32_16x16x32_bf16
    executed = (tile_taps * 4 * 8                          # conv taps: 4 k-steps x 8 channel tiles
                + ntiles * 3 * 8                           # stem, K = 96
                + ntiles * 4 * 8                           # final_conv.0
                + ntiles * 4) * mfma                       # final_conv.2 (one 16-row tile)
    dense = 2.0 * L * (45 * 128 + n * 9 * 128 * 128 + 128 * 128 + 128 * 5)
    return executed / dense


def sample_bf16(model, x, table, sigma_last, seed, sample_offset, uniforms):
    """Diffusion._sample on the fused kernel.  x: int64 [B,L] prior on the device."""
    bb = model.backbone
    B, L = x.shape
    if L > MAX_L:
        raise NotImplementedError(f'the fused bf16 loop holds one sequence per workgroup in LDS: L <= {MAX_L} '
                                  f"(got {L}); use hip_dtype='fp32'")
    dev = x.device
    num_steps = len(table)
    noise_removal = bool(model.config.sampling.noise_removal)
    # Schedule tensors live on the device across calls: a host-to-device copy from pageable memory waits for the
    # kernels already queued on the stream, which would serialise every call's host work behind the previous launch.
    key = (num_steps, table[0][1] if table else 0.0, table[-1][3] if table else 0.0, sigma_last, model.time_conditioning,
           bb._weights_key(), str(dev))
    cache = getattr(model, '_sched_cache', None)
    if cache is None or cache[0] != key:
        if model.time_conditioning:
            sig = [row[1] for row in table] + [sigma_last]
            tbias = bb.time_bias(torch.tensor(sig, dtype=torch.float32, device=dev))
            tb_stride = tbias.stride(0)
        else:
            tbias = bb.time_bias(torch.zeros(1, dtype=torch.float32, device=dev))
            tb_stride = 0
        mc = torch.tensor([[row[2], row[3]] for row in table], dtype=torch.float32, device=dev)
        model._sched_cache = cache = (key, tbias, tb_stride, mc)
    _, tbias, tb_stride, mc = cache
    u = None
    if uniforms is not None:
        u = torch.stack([uniforms(i) for i in range(num_steps)]).contiguous()
        assert u.shape == (num_steps, B, L, ops.V) and u.dtype == torch.float32
    out = torch.empty_like(x)
    _launch(bb, tbias, tb_stride, mc, u, x, out, None, -1, B, L, num_steps, noise_removal, seed, sample_offset)
    return out


def backbone_logits_bf16(bb, x, sigma_scalar=0.0, dbg_layer=None):
    """Raw logits [B,L,5] of ONE denoiser forward on the fused kernel's arithmetic (int tokens,
    one sigma for the whole batch).  Used by the parity tests of the bf16 path."""
    B, L = x.shape
    dev = x.device
    tbias = bb.time_bias(torch.full((1,), float(sigma_scalar), dtype=torch.float32, device=dev))
    logits = torch.empty(B, L, ops.V, dtype=torch.float32, device=dev)
    out = torch.empty_like(x)
    if dbg_layer is not None:
        feat = torch.zeros(B, L, ops.C, dtype=torch.float32, device=dev)
        _launch(bb, tbias, 0, None, None, x.contiguous(), out, logits, 0, B, L, 0, True, 0, 0, feat, dbg_layer)
        return logits, out, feat
    _launch(bb, tbias, 0, None, None, x.contiguous(), out, logits, 0, B, L, 0, True, 0, 0)
    return logits, out


def hard_steps_bf16(model, batch, table, n_hard, seed, sample_offset, uniforms):
    """The first `n_hard` reverse steps of a sampling run from the all-mask prior on the fused kernel (no noise removal).
    Returns (x after those steps int64 [B,L], trajectory int8 [n_hard,B,L] = the state each step started from)."""
    bb = model.backbone
    L = model.config.model.length
    if L > MAX_L:
        raise NotImplementedError(f'the fused bf16 loop holds L <= {MAX_L}')
    dev = model.device
    if model.time_conditioning:
        tbias = bb.time_bias(torch.tensor([row[1] for row in table[:n_hard]], dtype=torch.float32, device=dev))
        tb_stride = tbias.stride(0)
    else:
        tbias = bb.time_bias(torch.zeros(1, dtype=torch.float32, device=dev))
        tb_stride = 0
    mc = torch.tensor([[row[2], row[3]] for row in table[:n_hard]], dtype=torch.float32, device=dev)
    u = None
    if uniforms is not None:
        u = torch.stack([uniforms(i) for i in range(n_hard)]).contiguous()
    out = torch.empty(batch, L, dtype=torch.int64, device=dev)
    traj = torch.empty(n_hard, batch, L, dtype=torch.int8, device=dev)
    _launch(bb, tbias, tb_stride, mc, u, None, out, None, -1, batch, L, n_hard, False, seed, sample_offset, traj=traj)
    return out, traj


# ---- layer-per-launch bf16 path (any length) ------------------------------------------------------------------------
def _i64_ptr(t):
    if not t.is_cuda or t.dtype != torch.int64 or not t.is_contiguous():
        raise TypeError('expected a contiguous int64 tensor on the GPU')
    return ctypes.c_void_p(t.data_ptr())


def _bf16_ptr(t, name):
    if t is None:
        return None
    if not t.is_cuda or t.dtype != torch.bfloat16 or not t.is_contiguous():
        raise TypeError(f'{name}: expected a contiguous bfloat16 tensor on the GPU')
    return ctypes.c_void_p(t.data_ptr())


def forward_bf16_layers(bb, seq, tbias, tb_stride, work=None):
    """Raw logits [B,L,5] of the denoiser with one bf16-MFMA launch per residual layer (drk_layer_bf16): the path for
    sequences longer than the fused sampler's 224 positions.  Rounding points of the fused kernel in the residual stack
    and the first head conv (bf16 weights and LayerNorm outputs, fp32 accumulation and residual stream); stem and the
    128 -> 5 output conv run on the exact-fp32 kernels.  seq int64 [B,L] or float32 [B,L,5];
    tbias [nb, n_layers, 128] with tb_stride floats between batch rows (0: one row for the batch)."""
    pk, st = bb.packed(), _packed(bb)
    B, L = seq.shape[0], seq.shape[1]
    n = bb.num_layers
    if work is None:
        work = {}
    def buf(name, dtype, *shape):
        t = work.get(name)
        if t is None or t.shape != torch.Size(shape) or t.dtype != dtype:
            t = torch.empty(*shape, dtype=dtype, device=seq.device)
            work[name] = t
        return t
    h = [buf('h0', torch.bfloat16, B, L, ops.C), buf('h1', torch.bfloat16, B, L, ops.C)]
    fp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    vec = st['vecs']                               # [(n+1)*3, 128]: see pack_stream
    if seq.dtype == torch.int64:                   # tokens: stem and first LayerNorm in one launch
        feat, _ = ops.stem_ln_bf16(seq, pk['stem_w'], pk['stem_b'], tbias[0, 0], tb_stride, vec[1], vec[2],
                                   feat=buf('feat', torch.float32, B, L, ops.C), out=h[0])
    else:
        feat = ops.stem_fwd(seq, pk['stem_w'], pk['stem_b'], out=buf('feat', torch.float32, B, L, ops.C))
        _lib.call('drk_ln_fwd_bf16', fp(feat), fp(tbias[0, 0]), tb_stride, fp(vec[1]), fp(vec[2]), _bf16_ptr(h[0], 'h'), None, None,
                  B, L, ops._stream())
    flop = 2.0 * B * L * 9 * ops.C * ops.C
    for i in range(n):
        last = i + 1 == n
        w_i = st['wstream'][9 * i:]
        args = (_bf16_ptr(h[i & 1], 'h'), fp(feat), fp(feat), _bf16_ptr(h[(i + 1) & 1], 'h_out'), fp(w_i), fp(vec[3 * (i + 1)]),
                None if last else fp(tbias[0, i + 1]), tb_stride, None if last else fp(vec[3 * (i + 1) + 1]),
                None if last else fp(vec[3 * (i + 1) + 2]), B, L, 9, bb.dilation(i), 0 if last else 1, 1, None, None, None, ops._stream())
        ops.PROFILE.bracket('k_layer_bf16', flop, lambda: _lib.call('drk_layer_bf16', *args))
    # final_conv.0 (1x1) + ReLU + final_conv.2 (128 -> 5) in one launch: the 128-channel intermediate never leaves the chip
    logits = buf('logits', torch.float32, B, L, ops.V)
    _lib.call('drk_head_bf16', _bf16_ptr(h[n & 1], 'h'), fp(st['wstream'][9 * n:]), fp(vec[3 * n + 1]), fp(pk['head2_w']),
              fp(pk['head2_b']), fp(logits), B, L, ops._stream())
    return logits


# ---- several layers per launch (drk_layer_group_bf16): long sequences ---------------------------------------------------
GROUP_HALO_LAYERS = None   # dilation-1 layers per launch in halo mode (4 recomputed rows per layer and side) and
GROUP_HALO_TILES = None    # image rows / 16 there (14, 16 or 18); None: whatever _halo_shape finds cheapest for the length
GROUP_FUSE_ENDS = True     # the stem runs in the first launch (token input), the head in the last; False: their own launches
GROUP_MIN_FILL = 0.6       # the dispatcher below takes the groups when a stride-mode image is at least this full


def _halo_shape(L, run):
    """(layers per launch, image tiles) for a run of `run` dilation-1 layers at length L: the pair with the fewest
    16-row tile passes per layer, a launch's fixed cost (stream in and out, first LayerNorm: about 12 tile passes per
    workgroup) spread over its layers.  L = 1024: (4, 18) - 256 of 288 image rows written, 4 workgroups a sequence."""
    best = None
    for hl in (4, 3, 2, 1):
        if hl > run:
            continue
        for ht in (18, 16, 14):
            T = 16 * ht - 8 * hl
            cost = -(-L // T) * (ht + 12.0 / hl)
            if best is None or cost < best[0]:
                best = (cost, hl, ht)
    return best[1], best[2]


def group_plan(bb, L, halo_layers=None, halo_tiles=None):
    """How the residual stack splits into drk_layer_group_bf16 launches for length L: a list of
    (first_layer, n, stride, tiles16), or None when some layer fits neither mode.  Runs of dilation-1 layers go `halo_layers`
    at a time with a recomputed halo; a layer of dilation d > 1 opens a stride-d group that takes every following layer
    whose dilation is d x {1..4, 16} (the rows of one residue class mod d are closed under all of their shifts)."""
    dil = [bb.dilation(i) for i in range(bb.num_layers)]
    run = 0
    while run < len(dil) and dil[run] == 1:
        run += 1
    hl, ht = halo_layers or GROUP_HALO_LAYERS, halo_tiles or GROUP_HALO_TILES
    if hl is None or ht is None:
        auto = _halo_shape(L, max(run, 1))
        hl, ht = hl or auto[0], ht or auto[1]
    plan, i, n = [], 0, len(dil)
    while i < n:
        d = dil[i]
        if d == 1:
            k = i
            while k < n and k - i < hl and dil[k] == 1:
                k += 1
            if _lib.load().drk_layer_group_bf16_rows(k - i, ht) <= 0:
                return None
            plan.append((i, k - i, 1, ht))
        else:
            if -(-L // d) > 256:
                return None
            k = i
            while k < n and k - i < 12 and dil[k] % d == 0 and (dil[k] // d in (1, 2, 3, 4, 16)):
                k += 1
            plan.append((i, k - i, d, 16))
        i = k
    return plan


def group_launches(bb, seq, tbias, work, plan, keep_feat=False):
    """The argument tuples of the grouped forward's drk_layer_group_bf16 calls, in order: [(plan entry, args)], and whether
    the head still needs its own launch.  Buffers come from `work`."""
    pk, st = bb.packed(), _packed(bb)
    B, L = seq.shape[0], seq.shape[1]
    n = bb.num_layers
    def buf(name, dtype, *shape):
        t = work.get(name)
        if t is None or t.shape != torch.Size(shape) or t.dtype != dtype:
            t = torch.empty(*shape, dtype=dtype, device=seq.device)
            work[name] = t
        return t
    fp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    feat = [buf('feat', torch.float32, B, L, ops.C), buf('feat1', torch.float32, B, L, ops.C)]
    logits = buf('logits', torch.float32, B, L, ops.V)
    tokens = seq.dtype == torch.int64 and plan[0][0] == 0 and plan[0][2] == 1 and GROUP_FUSE_ENDS   # stem in the first launch
    tb = tbias[0].contiguous()
    cur, out, head = 0, [], False
    for gi, (l0, ng, stride, tiles16) in enumerate(plan):
        last = gi + 1 == len(plan)
        first = gi == 0 and tokens
        head = last and l0 + ng == n and GROUP_FUSE_ENDS and not keep_feat          # the head runs in the last launch
        dst = cur if (stride > 1 and not first) else cur ^ 1                        # stride mode works in place
        dil = (ctypes.c_int * ng)(*[bb.dilation(l0 + k) for k in range(ng)])
        fout = fp(feat[dst]) if (keep_feat or not last) else None
        hout = _bf16_ptr(buf('h0', torch.bfloat16, B, L, ops.C), 'h') if (last and not head) else None
        out.append(((l0, ng, stride, tiles16),
                    (None if first else fp(feat[cur]), _i64_ptr(seq) if first else None, fp(pk['stem_w']) if first else None,
                     fp(pk['stem_b']) if first else None, fout, hout, fp(logits) if head else None,
                     fp(pk['head2_w']) if head else None, fp(pk['head2_b']) if head else None, fp(st['wstream']), fp(st['vecs']),
                     fp(tb), B, L, l0, ng, stride, tiles16, dil, ops._stream())))
        cur = dst
    work['feat_final'] = feat[cur] if keep_feat else None
    return out, tokens, head


def forward_bf16_groups(bb, seq, tbias, work=None, plan=None, keep_feat=False):
    """Raw logits [B,L,5] as forward_bf16_layers gives them (same rounding points), with the residual layers run several
    per launch (drk_layer_group_bf16, csrc/drk_group_bf16.hip): only the fp32 residual stream passes through HBM between
    launches; with token input the stem runs inside the first launch, the head inside the last (GROUP_FUSE_ENDS).
    tbias [1, n_layers, 128]: one time for the batch.  keep_feat: the last group also writes its fp32 stream
    (work['feat_final']; tests), which nothing else reads; the head then takes its own launch."""
    pk, st = bb.packed(), _packed(bb)
    B, L = seq.shape[0], seq.shape[1]
    n = bb.num_layers
    if plan is None:
        plan = group_plan(bb, L)
    if plan is None:
        raise NotImplementedError(f'no group plan for length {L} with these dilations: use forward_bf16_layers')
    if tbias.shape[0] != 1:
        raise NotImplementedError('drk_layer_group_bf16 takes one time for the batch: use forward_bf16_layers')
    if work is None:
        work = {}
    launches, tokens, head = group_launches(bb, seq, tbias, work, plan, keep_feat)
    if not tokens:
        ops.stem_fwd(seq, pk['stem_w'], pk['stem_b'], out=work['feat'])
    for (l0, ng, stride, tiles16), args in launches:
        ops.PROFILE.bracket('k_group_bf16', 2.0 * B * L * 9 * ops.C * ops.C * ng, lambda: _lib.call('drk_layer_group_bf16', *args))
    if not head:
        fp = lambda t: ctypes.c_void_p(t.data_ptr())
        _lib.call('drk_head_bf16', _bf16_ptr(work['h0'], 'h'), fp(st['wstream'][9 * n:]), fp(st['vecs'][3 * n + 1]), fp(pk['head2_w']),
                  fp(pk['head2_b']), fp(work['logits']), B, L, ops._stream())
    return work['logits']


def forward_bf16_long(bb, seq, tbias, tb_stride, work=None):
    """The bf16 forward for sequences beyond the fused sampler: several layers per launch where the shape fills the
    group kernel's images (one time for the batch, ceil(L / stride) of 256 rows at least GROUP_MIN_FILL), else one launch
    per layer."""
    L = seq.shape[1]
    if tb_stride == 0 and tbias.shape[0] == 1 and getattr(bb, 'width', 128) == 128:
        key = (L, tuple(bb.dilation(i) for i in range(bb.num_layers)))
        cache = getattr(bb, '_group_plans', None)
        if cache is None:
            cache = bb._group_plans = {}
        if key not in cache:
            plan = group_plan(bb, L)
            if plan is not None and any(s > 1 and -(-L // s) < GROUP_MIN_FILL * 256 for _, _, s, _ in plan):
                plan = None
            cache[key] = plan
        if cache[key] is not None:
            return forward_bf16_groups(bb, seq, tbias, work, cache[key])
    return forward_bf16_layers(bb, seq, tbias, tb_stride, work)


# ---- the fused train kernels' lane-order saved blocks, mapped back onto rows (tests, gate logging) ----------------------
FEATN_SEQ = 8 * 14 * 256      # floats per sequence and layer in featn (drk_train_fused_bf16.hip)
MASKP_SEQ = 8 * 64            # uint64 per sequence and layer in maskp


def _native_maps(L, device):
    """For every (position p < L, channel c): the float offset inside a sequence's featn piece, and (word, bit) of its gate
    inside a sequence's maskp piece — the accumulator layout of the fused kernels (wave = ct + 4 grp; slot = 2 t + s; lane =
    16 g + n; position 16 (grp + 2 t) + (n ^ ((n & 8) >> 1)); channel 32 ct + 16 s + 4 g + j)."""
    p = torch.arange(L, device=device)[:, None]
    c = torch.arange(128, device=device)[None, :]
    T, pn = p >> 4, p & 15
    grp, t = T & 1, T >> 1
    n = pn ^ ((pn & 8) >> 1)
    ct, r = c >> 5, c & 31
    sgrp, g, j = r >> 4, (r & 15) >> 2, r & 3
    lane = 16 * g + n
    wv = ct + 4 * grp
    feat_off = ((wv * 14 + 2 * t + sgrp) * 64 + lane) * 4 + j
    word = wv * 64 + lane
    bit = 8 * t + 4 * sgrp + j
    return feat_off, word, bit


def saved_feats_rows(feats, B, L, n_layers, fused):
    """The forward's `feats` buffer (ops.denoiser_saved_buffers) as float32 [n_layers+1, B, L, 128]."""
    M = B * L
    if not fused:
        return feats[:(n_layers + 1) * M * 128].view(n_layers + 1, B, L, 128)
    off, _, _ = _native_maps(L, feats.device)
    featn = feats[M * 128:M * 128 + n_layers * B * FEATN_SEQ].view(n_layers, B, FEATN_SEQ)
    rows = featn[:, :, off.reshape(-1)].view(n_layers, B, L, 128)
    return torch.cat([feats[:M * 128].view(1, B, L, 128), rows])


def saved_masks_rows(masks, B, L, n_layers, fused):
    """The forward's `masks` buffer as uint8 [n_layers, B, L, 128] (1 = the ReLU gate was open)."""
    if not fused:
        return masks[:n_layers * B * L * 128].view(n_layers, B, L, 128)
    _, word, bit = _native_maps(L, masks.device)
    mp = masks[:n_layers * B * MASKP_SEQ * 8].view(torch.int64).view(n_layers, B, MASKP_SEQ)
    w = mp[:, :, word.reshape(-1)].view(n_layers, B, L, 128)
    return ((w >> bit) & 1).to(torch.uint8)
