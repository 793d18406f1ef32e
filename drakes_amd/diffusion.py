"""Drop-in for the sampling surface of the reference `Diffusion` class
(/root/reference/drakes_dna/diffusion_gosai_update.py:74-144, :243-275, :438-440, :462-486,
:506-630), running on hand-written gfx950 kernels.

Same constructor, method names (leading underscores included), argument meaning, return
shapes/dtypes and state_dict keys.  What is NOT here (pre-training hooks, EMA, the guided
baselines, the gReLU reward oracle) is listed in DESIGN.md "out of scope".

Differences a caller can see, all additive:
  * randomness: the reference calls torch.rand_like per step; here each step's uniforms are a
    counter-based function of (seed, step, global sample index, position, class).  `seed`
    defaults to a draw from torch's CPU generator, so `torch.manual_seed(s)` still makes a run
    reproducible.  `uniforms=` injects explicit noise (tests), `sample_offset=` is the global
    index of row 0 (data-parallel shards draw the noise of their own rows).
  * `hip_dtype` ('fp32' | 'bf16x3' | 'bf16') selects the exact-fp32 kernels (the parity path), the same step loop with
    the convolutions at fp32 grade on the bf16 matrix cores (three bf16 planes per operand: logits within ~1e-5 of the
    exact path's, ~3x its speed; under autograd the forward and input-gradient convolutions use it, weight gradients stay
    on the f32 MFMA), or the bf16-MFMA fast path (one launch per `_sample`; differentiable steps on the bf16 layer kernels);
    `hip_graph = True` replays the fp32 step loop from a captured hipGraph.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import noise_schedule, ops
from .models import dnaconv


class Diffusion(nn.Module):
    def __init__(self, config, eval=False):
        super().__init__()
        if eval:
            raise NotImplementedError('eval=True loads the gReLU reward oracle and the Gosai dataset '
                                      '(reference :133-139); both are outside the sampling hot path')
        self.config = config
        self.vocab_size = 4
        self.sampler = self.config.sampling.predictor
        self.mask_index = self.vocab_size
        self.vocab_size += 1
        self.parameterization = self.config.parameterization
        if self.config.backbone == 'cnn':
            self.backbone = dnaconv.CNNModel(self.config.model, alphabet_size=self.vocab_size, num_cls=3)
        else:
            raise ValueError(f'Unknown backbone: {self.config.backbone}')
        self.T = self.config.T
        self.subs_masking = getattr(self.config, 'subs_masking', False)
        self.noise = noise_schedule.get_noise(self.config)
        self.time_conditioning = self.config.time_conditioning
        self.neg_infinity = -1000000.0
        self.hip_dtype = 'fp32'
        self.hip_soft_dtype = None      # differentiable (Gumbel-softmax) steps: None = follow hip_dtype; 'fp32' | 'bf16'
        self.hip_graph = False          # fp32 path: replay the step loop from a captured hipGraph (see _sample_graph)
        self._graph = None
        self._validate_configuration()
        self._work = {}

    def _validate_configuration(self):
        assert self.parameterization == 'subs'

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def dtype(self):
        return torch.float32

    def _soft_dtype(self):
        """Arithmetic of the differentiable denoiser forwards / backwards (soft steps, forward() under grad)."""
        return self.hip_soft_dtype or self.hip_dtype

    def refresh_weights(self):
        """Call after writing parameters in a way torch's version counter does not see (`p.data.copy_(...)`,
        `p.data.mul_(...)` — e.g. the reference's EMA `copy_to` / `restore`, models/ema.py:62,87): drops the packed
        weight copies, the bf16 fragment stream, the schedule cache, the captured graph and the logits cache."""
        self.backbone.invalidate()
        self._sched_cache = None
        self._graph = None
        self._soft_logits_cache = {}

    # ------------------------------------------------------------------ forward
    def _process_sigma(self, sigma):
        """reference :255-264"""
        if sigma.ndim > 1:
            sigma = sigma.squeeze(-1)
        if not self.time_conditioning:
            sigma = torch.zeros_like(sigma)
        assert sigma.ndim == 1, sigma.shape
        return sigma

    def _subs_parameterization(self, logits, xt):
        """reference :243-253 (one fused kernel; writes into `logits` like the reference)."""
        xt = xt if xt.dtype == torch.int64 else xt.contiguous()
        if torch.is_grad_enabled() and logits.requires_grad:
            from .autograd import subs_with_grad
            return subs_with_grad(logits, xt)
        return ops.subs_fwd(logits, xt, out=logits)

    def forward(self, x, sigma):
        """Returns log score — reference :266-275."""
        uniform = getattr(sigma, '_drk_uniform', None)     # set by this package's samplers on the condt tensors they record:
        sigma = self._process_sigma(sigma)                 # every row holds this one value (known without a device read)
        if not self.time_conditioning or uniform is not None:
            sigma = sigma[:1]                              # one time bias for the batch: same values, one row of work
        self.backbone.train_dtype = self._soft_dtype()
        logits = self.backbone(x, sigma)
        return self._subs_parameterization(logits=logits, xt=x)

    # ------------------------------------------------------------------ sampling helpers
    def _sample_prior(self, *batch_dims):
        """reference :438-440"""
        return self.mask_index * torch.ones(*batch_dims, dtype=torch.int64)

    def _draw_seed(self):
        return int(torch.randint(0, 2 ** 62, (1,)).item())

    def _shared_time_bias(self, sigma_scalar):
        """[1, n_layers, 128] time biases for one sigma shared by the whole batch."""
        s = torch.full((1,), float(sigma_scalar), dtype=torch.float32, device=self.device)
        if not self.time_conditioning:
            s.zero_()
        return self.backbone.time_bias(s)

    def _ddpm_update(self, x, t, dt, return_process=False, *, seed=None, step=0, sample_offset=0, uniforms=None):
        """One hard reverse step — reference :462-486.  `t` [B,1]; rows may sit at different times (then the kernel
        reads one (move_chance_t, move_chance_s) pair per row)."""
        sigma_t, _ = self.noise(t)
        sigma_s, _ = self.noise(t - dt)
        if sigma_t.ndim > 1:
            sigma_t = sigma_t.squeeze(-1)
        if sigma_s.ndim > 1:
            sigma_s = sigma_s.squeeze(-1)
        move_chance_t = 1 - torch.exp(-sigma_t)
        move_chance_s = 1 - torch.exp(-sigma_s)
        mc_t, mc_s = float(move_chance_t[0]), float(move_chance_s[0])
        mc_rows = None
        if not (bool((move_chance_t == move_chance_t[0]).all()) and bool((move_chance_s == move_chance_s[0]).all())):
            mc_rows = torch.stack([move_chance_t, move_chance_s], dim=-1).to(device=x.device, dtype=torch.float32).contiguous()
        if seed is None and uniforms is None:
            seed = self._draw_seed()
        with torch.no_grad():
            logits = self.backbone(x, self._process_sigma(sigma_t).to(x.device))
            x_next = ops.reverse_step_hard(logits, x.contiguous(), mc_t, mc_s, seed or 0, step, sample_offset, uniforms,
                                           mc_rows=mc_rows)
        if return_process:
            copy_flag = (x != self.mask_index).to(x.dtype)
            return x_next, x, sigma_t, move_chance_t[:, None, None], copy_flag
        return x_next

    @torch.no_grad()
    def _sample(self, num_steps=None, eps=1e-5, eval_sp_size=None, *, seed=None, sample_offset=0, uniforms=None):
        """Generate samples from the model — reference :506-550 (ddpm predictor).

        uniforms: optional callable step -> float32 [B,L,5] device tensor replacing the in-kernel
        Philox draw (noise injection for parity tests)."""
        if eval_sp_size is None:
            batch = self.config.loader.eval_batch_size
        else:
            batch = eval_sp_size
        if self.sampler != 'ddpm':
            raise NotImplementedError(f"sampling.predictor={self.sampler!r}: only 'ddpm' is on the hot path")
        if num_steps is None:
            num_steps = self.config.sampling.steps
        L = self.config.model.length
        dev = self.device
        if seed is None:
            seed = self._draw_seed()
        table, sigma_last, _, _ = noise_schedule.step_table(self.noise, num_steps, eps)
        table = table.tolist()
        # _sample_prior (:438-440) built on the device: a host-to-device copy would wait for every kernel queued before it
        x = torch.full((batch, L), self.mask_index, dtype=torch.int64, device=dev)
        if batch == 0:
            return x
        bb = self.backbone
        fwd = bb.forward_x3 if self.hip_dtype == 'bf16x3' else bb.forward_f32
        if self.hip_dtype == 'bf16':
            from . import fastpath
            if L <= fastpath.MAX_L:
                return fastpath.sample_bf16(self, x, table, float(sigma_last), seed, sample_offset, uniforms)
            # longer sequences do not fit one workgroup's LDS: several residual layers per launch on 256-row images
            # (or one launch per layer where the shape does not fill them), same step kernels
            fwd = lambda seq, tb, stride, work: fastpath.forward_bf16_long(bb, seq, tb, stride, work)
        shared = None if self.time_conditioning else self._shared_time_bias(0.0)
        if self.hip_graph and uniforms is None and shared is not None and num_steps > 0:
            x = self._sample_graph(x, table, shared, seed, sample_offset, fwd)
            num_steps = 0                      # the loop below is skipped; noise removal follows
        for i in range(num_steps):
            _, sigma_t, mc_t, mc_s = table[i]
            tb = shared if shared is not None else self._shared_time_bias(sigma_t)
            logits = fwd(x, tb, 0, self._work)
            u = uniforms(i) if uniforms is not None else None
            ops.reverse_step_hard(logits, x, mc_t, mc_s, seed, i, sample_offset, u, out=x)
        if self.config.sampling.noise_removal:
            tb = shared if shared is not None else self._shared_time_bias(float(sigma_last))
            logits = fwd(x, tb, 0, self._work)
            x = ops.denoise_argmax(logits, x)
        return x

    def _sample_graph(self, x, table, shared, seed, sample_offset, fwd):
        """The step loop of _sample as ONE captured launch sequence (hipGraph), replayed once per step: the launches of
        a step (denoiser forward + fused reverse step) are recorded once per (batch, length); what changes from step to
        step or call to call (move chances, the Philox step index, seed) is read from device memory by the step kernel.  Removes the
        per-launch host cost (147 launches per step) at small batches.  Not with time conditioning or injected noise."""
        B, L = x.shape
        num_steps = len(table)
        dev = x.device
        mc = torch.tensor([[row[2], row[3]] for row in table], dtype=torch.float32, device=dev)
        # the graph owns every buffer it points at: the time bias is COPIED into st['shared'] before each replay, so a
        # cache hit does not depend on the allocator handing `shared` the same address again
        key = (B, L, num_steps, tuple(shared.shape), self.backbone._weights_key(), self.hip_dtype)
        st = self._graph
        if st is None or st['key'] != key:
            st = {'key': key, 'x': torch.empty_like(x), 'counter': torch.zeros(1, dtype=torch.int32, device=dev), 'mc': mc,
                  'rng': torch.zeros(2, dtype=torch.int64, device=dev), 'shared': shared.clone()}
            st['x'].copy_(x)
            work = {}
            self.backbone.packed()
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):          # warm-up outside capture: allocates the work buffers
                logits = fwd(st['x'], st['shared'], 0, work)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                logits = fwd(st['x'], st['shared'], 0, work)
                ops.reverse_step_hard_table(logits, st['x'], st['mc'], st['counter'], st['rng'], out=st['x'])
            st['graph'], st['work'] = graph, work
            self._graph = st
        st['shared'].copy_(shared)
        st['x'].copy_(x)
        st['mc'].copy_(mc)
        st['counter'].zero_()
        as_i64 = lambda v: v - (1 << 64) if v >= (1 << 63) else v          # uint64 bit pattern in an int64 tensor
        st['rng'].copy_(torch.tensor([as_i64(seed), as_i64(sample_offset)], dtype=torch.int64))
        for _ in range(num_steps):
            st['graph'].replay()
        return st['x'].clone()

    # ------------------------------------------------------------------ likelihood
    def q_xt(self, x, move_chance, *, seed=None, step=0, sample_offset=0, uniforms=None):
        """Computes the noisy sample xt — reference :425-436.  move_chance: [B,1] (one value per row) or a float."""
        mc, mc_rows = move_chance, None
        if torch.is_tensor(mc):
            if mc.numel() > 1 and not bool((mc == mc.reshape(-1)[0]).all()):
                mc_rows = mc.reshape(-1, 1).to(device=x.device, dtype=torch.float32).contiguous()
            mc = float(mc.reshape(-1)[0])
        if seed is None and uniforms is None:
            seed = self._draw_seed()
        return ops.q_xt(x.contiguous(), mc, seed or 0, step, sample_offset, uniforms, mc_rows=mc_rows)

    @torch.no_grad()
    def get_likelihood(self, x0, num_steps=None, eps=1e-5, n_samples=1, *, seed=None, sample_offset=0, uniforms=None):
        """Compute the likelihood of a sequence under the model — reference :885-926.

        x0: int64 [B, L] on the device.  uniforms: optional callable (sample, step) -> float32 [B,L] replacing the
        in-kernel Philox draw (sample s uses seed + s)."""
        if num_steps is None:
            num_steps = self.config.sampling.steps
        if seed is None:
            seed = self._draw_seed()
        x0 = x0.contiguous()
        B, L = x0.shape
        table, _, _, _ = noise_schedule.step_table(self.noise, num_steps, eps)
        table = table.tolist()
        bb = self.backbone
        fwd = bb.forward_x3 if self.hip_dtype == 'bf16x3' else bb.forward_f32
        if self.hip_dtype == 'bf16':       # batched forwards on the bf16 layer kernels (logits within the bf16 bar)
            from . import fastpath
            fwd = lambda seq, tb, stride, work: fastpath.forward_bf16_long(bb, seq, tb, stride, work)
        shared = None if self.time_conditioning else self._shared_time_bias(0.0)
        total = torch.zeros(B, dtype=torch.float32, device=x0.device)
        acc = torch.empty(B, L, dtype=torch.float32, device=x0.device)
        xt = torch.empty_like(x0)
        for s in range(n_samples):
            acc.zero_()
            for i in range(num_steps):
                _, sigma_t, mc_t, mc_s = table[i]
                # (move_chance_t - move_chance_s) / move_chance_t in fp32, as the reference computes it (:908)
                mult = float((torch.tensor(mc_t, dtype=torch.float32) - torch.tensor(mc_s, dtype=torch.float32))
                             / torch.tensor(mc_t, dtype=torch.float32))
                u = uniforms(s, i) if uniforms is not None else None
                ops.q_xt(x0, mc_t, seed + s, i, sample_offset, u, out=xt)
                tb = shared if shared is not None else self._shared_time_bias(sigma_t)
                logits = fwd(xt, tb, 0, self._work)
                ops.likelihood_accum(logits, x0, xt, mult, acc)
            total += acc.sum(dim=-1)
        return total / n_samples

    # ------------------------------------------------------------------ fine-tuning sampler
    def _ddpm_update_finetune_gradient(self, x, t, dt, copy_flag_temp, return_process=False, **kw):
        from . import finetune
        return finetune.ddpm_update_finetune_gradient(self, x, t, dt, copy_flag_temp, return_process, **kw)

    def _sample_finetune_gradient(self, num_steps=None, eps=1e-5, eval_sp_size=None, copy_flag_temp=None, **kw):
        """reference :589-630"""
        from . import finetune
        return finetune.sample_finetune_gradient(self, num_steps, eps, eval_sp_size, copy_flag_temp, **kw)

    # north_star spells the method without the underscore; harmless alias
    sample_finetune_gradient = _sample_finetune_gradient
