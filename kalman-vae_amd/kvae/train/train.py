"""Training step of the KVAE on MI355X: the body of the reference's train_one_epoch
(kvae/train/train.py:32-62 there: zero_grad, forward, compute_loss, backward, clip_grad_norm_(10),
Adam.step) and the three training phases of its main loop (train.py:142-207, 246-260) restated for one process per GPU.

  * gradients of all parameters live in ONE flat fp32 buffer (each p.grad is a view), so gradient
    clipping is two kernels and data-parallel training needs exactly ONE RCCL all-reduce per step
    (~0.45 MB, latency-bound on xGMI: a single bucket, no overlap machinery);
  * the whole step (the hand-written VAE kernels of csrc/vae_*.h for the reference's default shapes - MIOpen for
    other shapes -, the HIP LSTM / bi-GRU, the HIP LGSSM chain, loss, backward, clip, fused Adam) is captured into a
    hipGraph (torch.cuda.CUDAGraph) and replayed; with >1 rank the graph is cut around the all-reduce (or, with
    `graph_allreduce=True`, the RCCL call is captured too and the step is one graph);
  * no host synchronisation inside a step: losses stay on the device until the caller reads them
    (the reference forces six device->host syncs per step);
  * everything a schedule moves between steps lives in a DEVICE scalar the captured kernels read: beta of the KL
    term, the learning rate (Adam is built with a tensor lr, so torch's LRScheduler updates it in place), the
    Gumbel-softmax temperature tau (switch_dyn_param.py: the `tau` property) and the two loss weights kf_weight /
    vae_weight (the reference's phases set kf_weight = 0 while the VAE is pre-trained);
  * training phases: `set_training_phase` toggles requires_grad exactly as the reference does; a frozen parameter has
    no gradient, and kvae_clip_adam skips its slot the way clip_grad_norm_ / Adam skip `grad is None` - no moment, step-count
    or value update (torch's Adam counts steps per parameter, so do the slots).  A phase switch re-captures once.
Data parallelism: the ELBO is normalised by the LOCAL count of observed frames (kalman_filter.py:392 /
losses.py:82), so the global-batch gradient is sum_r(count_r * grad_r) / sum_r(count_r): each rank scales its flat
gradient by its own count, the count rides in one extra slot of the same buffer, and ONE sum-all-reduce carries both
(SURVEY.md section 5).  With mask == 1 and equal shards this is the plain mean.
"""
import gc
import os
import weakref
from contextlib import contextmanager

# hipBLASLt aborts the process when one of its calls lands inside hipGraph capture ("operation not
# permitted when stream is capturing", hipblaslt.cpp:171 on ROCm 7.2): route the few small GEMMs of the
# model (fc layers, LSTM head) through rocBLAS instead.  Must be set before the first addmm runs.
os.environ.setdefault("DISABLE_ADDMM_CUDA_LT", "1")
os.environ.setdefault("TORCH_BLAS_PREFER_HIPBLASLT", "0")

import torch
import torch.distributed as dist

PHASES = ("vae", "warmup", "all")


def _alpha_net_parameters(dyn):
    """The parameters the reference keeps frozen in "vae" and "warmup" (train.py:159-174, 190-205): the regime posterior of
    the switching model, lstm / mlp / head_w of the mixture model (K > 1 only)."""
    if getattr(dyn, "is_switching_dynamics", False):
        return list(dyn.markov_regime_posterior.parameters())
    out = []
    if dyn.K > 1:
        for name in ("lstm", "mlp", "head_w"):
            if hasattr(dyn, name):
                out += list(getattr(dyn, name).parameters())
    return out


def set_training_phase(model, phase: str):
    """requires_grad of every parameter for one of the reference's three phases (kvae/train/train.py:142-207 there):
      "vae"     encoder + decoder train; A, B, C, (Q) and the alpha-network / regime posterior are frozen;
      "warmup"  encoder + decoder + A, B, C (and Q of the switching model) train; the alpha-network stays frozen;
      "all"     everything trains."""
    assert phase in PHASES, phase
    for p in model.parameters():
        p.requires_grad = False
    dyn = model.kalman_filter.dyn_params
    if phase == "all":
        for p in model.parameters():
            p.requires_grad = True
        return
    for mod in (model.encoder, model.decoder):
        for p in mod.parameters():
            p.requires_grad = True
    if phase == "warmup":
        dyn.A.requires_grad = True
        dyn.B.requires_grad = True
        if hasattr(dyn, "Q"):
            dyn.Q.requires_grad = True
        dyn.C.requires_grad = True
        for p in _alpha_net_parameters(dyn):
            p.requires_grad = False


def phase_for_epoch(epoch, pretrain_vae_epochs=5, warmup_epochs=10):
    """(phase, kf_weight, vae_weight) of a 1-based epoch, as the reference's main loop picks them (train.py:246-260;
    the defaults are TrainingConfig's, train.py:353-354)."""
    if epoch <= pretrain_vae_epochs:
        return "vae", 0.0, 1.0
    if epoch <= pretrain_vae_epochs + warmup_epochs:
        return "warmup", 1.0, 1.0
    return "all", 1.0, 1.0


class Trainer:
    def __init__(self, model, lr=7e-3, weight_decay=0.0, grad_clip_norm=10.0, kf_weight=1.0, vae_weight=1.0,
                 use_graph=True, world_size=1, overlap_lgssm=True, reference_logging=False, graph_allreduce=False):
        """reference_logging: also compute what the reference's step computes for logging only - sigmoid(x_logits)
        (model.py:165-168 there) and the active-unit statistics (model.py:229) - as device tensors in `self.out`.
        graph_allreduce: with world_size > 1, capture the RCCL all-reduce into the step's hipGraph (one replay per step)
        instead of cutting the graph around an eager call."""
        self.model, self.clip = model, grad_clip_norm
        self.world = world_size
        self.reference_logging = bool(reference_logging)
        # ALL parameters, as the reference's Adam(model.parameters()) (train.py:236): which of them a step updates is
        # decided per step by who has a gradient (the phases), not at construction
        self.params = list(model.parameters())
        dev = self.params[0].device
        on_gpu = dev.type == "cuda"
        n_par = sum(p.numel() for p in self.params)
        self._flat = torch.zeros(n_par + 1, device=dev, dtype=torch.float32)   # + 1: the local observed-frame count
        self.flat_grad, self._count = self._flat[:n_par], self._flat[n_par:]
        off, self.grad_views = 0, []
        for p in self.params:
            self.grad_views.append(self.flat_grad[off:off + p.numel()].view_as(p))
            p.grad = self.grad_views[-1]
            off += p.numel()
        # a TENSOR learning rate: the fused capturable Adam kernel reads it from memory, and torch's LRScheduler
        # updates it with fill_() - the decay of train.py:268-269 then reaches a captured step (a float would be baked in)
        self.lr_t = torch.tensor(float(lr), device=dev, dtype=torch.float32) if on_gpu else float(lr)
        self.opt = torch.optim.Adam(self.params, lr=self.lr_t, weight_decay=weight_decay, capturable=on_gpu, fused=on_gpu)
        self._flat_step = (on_gpu and all(p.dtype == torch.float32 for p in self.params)
                           and os.environ.get("KVAE_FLAT_ADAM", "1") != "0")   # 0: torch's fused Adam + aten clip (A/B runs)
        self._active_host = None
        if self._flat_step:
            prev = getattr(model, "_flat_trainer", None)
            prev = prev() if prev is not None else None
            if prev is not None and prev is not self:
                prev._release()   # its captured graphs would keep updating storage the parameters no longer live in
            self._flatten_optimizer(n_par, dev)
            # (a weak method: the optimizer must not keep its Trainer - and the Trainer's captured graphs - alive in a reference
            # cycle that only the cyclic collector frees, possibly in the middle of a later capture)
            relink = weakref.WeakMethod(self.relink_optimizer_state)
            self.opt._kvae_relink = lambda: relink() and relink()()
            model._flat_trainer = weakref.ref(self)
        self._released = False
        dyn = model.kalman_filter.dyn_params
        if on_gpu and hasattr(dyn, "tau_scalar"):
            dyn.tau_scalar(dev)   # create the device scalar of tau outside any capture
        # beta of the KL term and the two loss weights live in device scalars: schedules / phases move them without re-capture
        self.beta_t = torch.tensor(float(model.beta), device=dev, dtype=torch.float32)
        model.beta = self.beta_t
        self._weights_t = torch.tensor([float(vae_weight), float(kf_weight)], device=dev, dtype=torch.float32)
        self._kf_weight, self._vae_weight = float(kf_weight), float(vae_weight)
        self.phase = None
        self.use_graph = bool(use_graph) and on_gpu
        self.graph_allreduce = bool(graph_allreduce)
        # the LGSSM chain runs on its own stream next to the decoder convolutions (fork/join inside the graph) ...
        self.lgssm_stream = torch.cuda.Stream() if (on_gpu and overlap_lgssm and self.use_graph) else None
        if self.lgssm_stream is not None and hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
            # gradients of the LGSSM parameters are produced on the side stream by design
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        # ... and runs its backward right behind its forward (KVAE.early_kf_backward), not when loss.backward() reaches it.
        # Both are properties of THIS trainer's step: the model carries them only while _forward_backward runs.
        self.early_kf_backward = self.lgssm_stream is not None and os.environ.get("KVAE_EARLY_KF_BWD", "1") != "0"
        self._graphs = {}          # (x.shape, mask is None) -> captured step
        self.captures = 0          # hipGraph captures so far (one per batch shape and per phase)
        self.graph_fb = self.graph_opt = None      # the most recently replayed pair (graph_opt: multi-rank cut only)
        self.static_x = self.static_mask = None
        self.out = {}

    # -- loss weights and phases ---------------------------------------------------------------------
    @property
    def kf_weight(self):
        return self._kf_weight

    @property
    def vae_weight(self):
        return self._vae_weight

    def set_loss_weights(self, kf_weight=None, vae_weight=None):
        """kf_weight / vae_weight of compute_loss (train.py:246-260 there).  Device scalars: a captured step follows them.
        Only a change of kf_weight to or from exactly 0 can change what a step launches (see _kf_value_only)."""
        was = self._kf_value_only()
        if kf_weight is not None:
            self._kf_weight = float(kf_weight)
        if vae_weight is not None:
            self._vae_weight = float(vae_weight)
        self._weights_t.copy_(torch.tensor([self._vae_weight, self._kf_weight], dtype=torch.float32), non_blocking=False)
        if was != self._kf_value_only():
            self._drop_graphs()

    def set_training_phase(self, phase, kf_weight=None, vae_weight=None):
        """The reference's set_training_phase(model, phase) plus the loss weights its main loop pairs with the phase
        (kf_weight 0 in "vae", else 1; vae_weight 1) unless given.  The next step() re-captures."""
        set_training_phase(self.model, phase)
        self.phase = phase
        self._drop_graphs()
        self.set_loss_weights(kf_weight=(0.0 if phase == "vae" else 1.0) if kf_weight is None else kf_weight,
                              vae_weight=1.0 if vae_weight is None else vae_weight)

    def _kf_value_only(self):
        """True when the LGSSM term cannot move anything: its weight is exactly 0 (so d loss / d a through it is an exact
        zero) and none of its parameters takes a gradient (the "vae" phase).  The chain's forward then runs without a tape,
        for the logged elbo_kf only, and its backward is not launched at all."""
        return self._kf_weight == 0.0 and not any(p.requires_grad for p in self.model.kalman_filter.parameters())

    def _drop_graphs(self):
        self._graphs.clear()
        self.graph_fb = self.graph_opt = None
        self.static_x = self.static_mask = None

    def _release(self):
        """Another Trainer took over the model's parameters (their storage moved into ITS flat buffers)."""
        self._released = True
        self._drop_graphs()

    @contextmanager
    def _schedule(self):
        """The model carries this trainer's step schedule (side stream, early LGSSM backward, value-only LGSSM term) only
        inside the trainer's own forward+backward: a plain model(x) / compute_loss / backward outside is the reference's."""
        m = self.model
        saved = (m.lgssm_stream, m.early_kf_backward, m.kf_value_only)
        value_only = self._kf_value_only()
        m.lgssm_stream = self.lgssm_stream
        m.early_kf_backward = self.early_kf_backward and not value_only
        m.kf_value_only = value_only
        try:
            yield
        finally:
            m.lgssm_stream, m.early_kf_backward, m.kf_value_only = saved

    # -- the three segments of a step ---------------------------------------------------------------
    def _forward_backward(self, x, mask=None):
        """mask None == all frames observed (the reference passes a mask of ones, train.py:41)."""
        for p in self.params:   # autograd then hands over each gradient tensor as is (no per-parameter add kernel)
            p.grad = None
        self.model.kalman_filter.dyn_params.reset_state()
        with self._schedule():
            outputs = self.model(x, mask=mask, with_recon=self.reference_logging)
            losses = self.model.compute_loss(x, outputs, kf_weight=self._kf_weight, vae_weight=self._vae_weight, mask=mask,
                                             with_metrics="device" if self.reference_logging else False,
                                             weights_dev=self._weights_t if x.is_cuda else None)
            losses["loss"].backward()
        self._gather_grads()
        self.out = {k: losses[k].detach() for k in ("loss", "elbo_kf", "elbo_vae_total")}
        if self.reference_logging:
            self.out.update(x_recon=outputs["x_recon"], active_units=losses["active_units"],
                            latent_variances=losses["latent_variances"])
        if self.world > 1:   # weight of this rank's gradient in the global-batch gradient (module docstring)
            if mask is None:
                self._count.fill_(float(x.shape[0] * x.shape[1]))
            else:
                self._count.copy_(mask.sum().reshape(1))
            self.flat_grad.mul_(self._count.clamp(min=1.0))

    def _gather_grads(self):
        """All gradients into the flat buffer with ONE multi-tensor copy.  A parameter without a gradient (frozen by the
        phase) is an inactive slot: zeros in the flat buffer (so the all-reduce and the norm see nothing), skipped by the
        optimizer.  With the flat optimizer p.grad becomes the flat view again; with torch's Adam it stays None, which is
        how that optimizer knows to skip it."""
        got = [(v, p.grad) for v, p in zip(self.grad_views, self.params) if p.grad is not None]
        active = tuple(p.grad is not None for p in self.params)
        if got:
            torch._foreach_copy_([v for v, _ in got], [g for _, g in got])
        for v, p, on in zip(self.grad_views, self.params, active):
            if not on:
                v.zero_()
            p.grad = v if (on or self._flat_step) else None
        if self._flat_step and active != self._active_host:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("the set of parameters with a gradient changed inside hipGraph capture")
            self._active_host = active
            self._seg_active.copy_(torch.tensor(active, dtype=torch.float32))

    def _allreduce(self):
        if self.world > 1:
            dist.all_reduce(self._flat)   # sum of count-weighted gradients and, in the last slot, of the counts

    def _clip_and_update(self):
        if self._flat_step:
            assert len(self.opt.param_groups) == 1, "the flat optimizer step serves the Trainer's own single parameter group"
            self._clip_and_update_flat()      # (divides by the all-reduced frame count itself when world > 1)
            return
        if self.world > 1:
            self.flat_grad.div_(self._count.clamp(min=1.0))
        self._clip_and_update_local()

    # -- clip + Adam as two launches on flat buffers (GPU) --------------------------------------------------
    def _flatten_optimizer(self, n_par, dev):
        """Parameters, exp_avg and exp_avg_sq become views of three flat buffers, so that clip_grad_norm_ + Adam.step are
        kvae_clip_adam's two launches instead of a dozen (norm, clamp, reciprocal, scale, the foreach kernels of the fused
        Adam).  `self.opt` stays a torch.optim.Adam whose state ENTRIES are those views (and one step counter per
        parameter): state_dict() / LR schedulers / the reference-format checkpoint see an ordinary Adam; only step() is never
        called on it."""
        self._flat_p = torch.empty(n_par, device=dev, dtype=torch.float32)
        self._flat_m = torch.zeros(n_par, device=dev, dtype=torch.float32)
        self._flat_v = torch.zeros(n_par, device=dev, dtype=torch.float32)
        n_seg = len(self.params)
        self._seg_steps = torch.zeros(n_seg, device=dev, dtype=torch.float32)    # torch's Adam: one `step` per parameter
        self._seg_active = torch.ones(n_seg, device=dev, dtype=torch.float32)    # 0: frozen (no gradient this phase)
        self._seg_of = torch.repeat_interleave(torch.arange(n_seg, dtype=torch.int32),
                                               torch.tensor([p.numel() for p in self.params])).to(dev)
        self._norm_t = torch.zeros((), device=dev, dtype=torch.float32)
        self._ca_ws = torch.empty(1024, device=dev, dtype=torch.float32)
        off = 0
        with torch.no_grad():
            for p in self.params:
                n = p.numel()
                view = self._flat_p[off:off + n].view_as(p)
                view.copy_(p)
                p.data = view
                off += n
        self._link_state()

    def _link_state(self, take_values=False):
        """(Re)point self.opt.state at the flat moment buffers; take_values: first copy what the state currently holds
        (after Optimizer.load_state_dict, which replaces the tensors) into them - a parameter the file has no state for
        (never updated: frozen in every epoch before the save) restarts from zero moments and step 0."""
        off = 0
        with torch.no_grad():
            for i, p in enumerate(self.params):
                n = p.numel()
                m, v = self._flat_m[off:off + n].view_as(p), self._flat_v[off:off + n].view_as(p)
                st = self.opt.state.get(p)
                if take_values:
                    if st:
                        m.copy_(st["exp_avg"]), v.copy_(st["exp_avg_sq"])
                        self._seg_steps[i] = float(st["step"])
                    else:
                        m.zero_(), v.zero_()
                        self._seg_steps[i] = 0.0
                self.opt.state[p] = {"step": self._seg_steps[i], "exp_avg": m, "exp_avg_sq": v}
                off += n

    def relink_optimizer_state(self):
        """Call after self.opt.load_state_dict(...) (kvae.train.checkpoint.load_checkpoint does)."""
        if self._flat_step:
            self._link_state(take_values=True)

    def _clip_and_update_flat(self):
        from .. import _native
        g = self.opt.param_groups[0]
        lib = _native.lib_for(self._flat_p)
        lr = g["lr"]
        b1, b2 = g["betas"]
        lib.check(lib.dll.kvae_clip_adam(
            _native.ptr(self._flat_p), _native.ptr(self.flat_grad), _native.ptr(self._flat_m), _native.ptr(self._flat_v),
            self._flat_p.numel(), _native.ptr(self._seg_of), len(self.params), _native.ptr(self._seg_active),
            _native.ptr(self._seg_steps), _native.ptr(lr) if isinstance(lr, torch.Tensor) else None,
            0.0 if isinstance(lr, torch.Tensor) else float(lr), float(b1), float(b2), float(g["eps"]),
            float(g["weight_decay"]), float(self.clip or 0.0), _native.ptr(self._count) if self.world > 1 else None,
            _native.ptr(self._norm_t), _native.ptr(self._ca_ws), _native.stream_for(self._flat_p)), "kvae_clip_adam")
        if self.clip and self.clip > 0:
            self.out["grad_norm"] = self._norm_t
        self.opt._opt_called = True   # what Optimizer.step's wrapper records; LRScheduler.step() checks it to warn about call order

    def _clip_and_update_local(self):
        if self.clip and self.clip > 0:   # torch.nn.utils.clip_grad_norm_ on the flat view of all grads (frozen slots hold zeros)
            total = torch.linalg.vector_norm(self.flat_grad)
            self.flat_grad.mul_(torch.clamp(self.clip / (total + 1e-6), max=1.0))
            self.out["grad_norm"] = total
        self.opt.step()

    def set_beta(self, value: float):
        """Linear KL warm-up (reference train.py:30): updates the device scalar the captured graph reads."""
        self.beta_t.fill_(float(value))

    # -- public ---------------------------------------------------------------------------------------
    def set_lr(self, value: float):
        """Learning rate of every parameter group (the device scalar the captured Adam kernel reads)."""
        for g in self.opt.param_groups:
            if isinstance(g["lr"], torch.Tensor):
                g["lr"].fill_(float(value))
            else:
                g["lr"] = float(value)

    def set_tau(self, value: float):
        """Gumbel-softmax temperature of the switching dynamics (reference train.py:270-274)."""
        self.model.kalman_filter.dyn_params.tau = float(value)

    def step(self, x, mask=None):
        """One optimisation step on batch x [B,T,C,H,W] (already on the device); mask [B,T] (1 = observed) or None
        for all frames observed.  Returns device scalars.  A batch of another shape (the last partial batch of an epoch)
        gets a captured step of its own; the two most recent shapes are kept."""
        if self._released:
            raise RuntimeError("this Trainer was superseded: another Trainer was built on the same model and owns its "
                               "parameter storage now")
        if not self.use_graph:
            self._forward_backward(x, mask)
            self._allreduce()
            self._clip_and_update()
            return self.out
        key = (tuple(x.shape), mask is None)
        g = self._graphs.get(key)
        if g is None:
            if len(self._graphs) >= 2:
                self._graphs.pop(next(iter(self._graphs)))
            g = self._graphs[key] = self._capture(x, mask)
        self.graph_fb, self.graph_opt, self.static_x, self.static_mask, self.out = g["fb"], g["opt"], g["x"], g["mask"], g["out"]
        self.static_x.copy_(x, non_blocking=True)
        if mask is not None:
            self.static_mask.copy_(mask, non_blocking=True)
        self.graph_fb.replay()
        if self.graph_opt is not None:
            self._allreduce()
            self.graph_opt.replay()
        return self.out

    def _snapshot(self):
        """Parameters and optimizer state before the capture warm-up (which runs real optimizer steps so that the Adam
        moments exist at static addresses before capture)."""
        return ([p.detach().clone() for p in self.params],
                {id(p): {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in st.items()}
                 for p, st in self.opt.state.items()})

    @torch.no_grad()
    def _restore(self, snap):
        saved_p, saved_state = snap
        torch._foreach_copy_(self.params, saved_p)
        for p, st in self.opt.state.items():
            old = saved_state.get(id(p))
            for k, v in st.items():
                if not isinstance(v, torch.Tensor):
                    if old is not None:
                        st[k] = old[k]
                elif old is not None:
                    v.copy_(old[k])
                else:
                    v.zero_()      # state created by the warm-up: back to a fresh optimizer (step 0, zero moments), in place

    def _decoder_workgroups(self, frames):
        """Persistent decoder-block workgroups for the captured step.  The switching model's side chain (bi-GRU, regime chain,
        LGSSM, all above the 32 registers per lane the Winograd workgroups leave free) needs CUs of its own while it is what
        the encoder's backward waits for.  Round 2 found that up to 256 sequences of T = 50; with round 3's side chain (lane-grid
        regime chain, no library GEMM) the frame pass of configs[1] hides it again and every CU taken from the decoder costs
        (12800 frames: 2.847 / 2.860 / 2.884 / 2.886 ms at 256 / 248 / 240 / 224 workgroups), while below that the chain is
        still the critical path (6400 frames: 1.760 at 224 against 1.799; 3200: 1.212 against 1.250; configs[3]: neutral).
        The lstm model's chain is hidden from 6400 frames on (1.561 / 1.562 / 1.593 ms at 256 / 240 / 224 workgroups there,
        2.110 / 2.161 / 2.208 at 9600); at 3200 frames it is not (configs[3] with the LSTM alpha-net, T = 100: 1.144 -> 1.118 ms at
        224; T = 50: 1.040 -> 1.028)."""
        dyn = self.model.kalman_filter.dyn_params
        if self.lgssm_stream is None or self._kf_value_only():
            return 256
        limit = 12800 if getattr(dyn, "is_switching_dynamics", False) else 4800
        return 224 if frames < limit else 256

    def _capture(self, x, mask=None):
        from .. import _native
        lib = _native.lib_for(x)
        prev = lib.dll.kvae_dec_up_set_workgroups(self._decoder_workgroups(x.shape[0] * x.shape[1]))
        # An object the cyclic collector happens to free DURING capture (an older Trainer's hipGraph, a tensor of its pool) makes
        # HIP calls that are illegal while a stream captures, and the process aborts: collect now, and not again until done.
        self.captures += 1
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            return self._capture_graphs(x, mask)
        finally:   # process-wide setting: only the launches recorded above are meant
            if gc_was_on:
                gc.enable()
            lib.dll.kvae_dec_up_set_workgroups(prev if prev != 256 else 0)

    def _capture_graphs(self, x, mask=None):
        try:
            torch.backends.cuda.preferred_blas_library("cublas")   # == rocBLAS on ROCm (see module header)
        except Exception:
            pass
        static_x = x.clone()
        static_mask = None if mask is None else mask.to(device=x.device, dtype=torch.float32).clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):          # warm-up outside capture (MIOpen find, allocator, Adam state)
            snap = self._snapshot()
            for _ in range(3):
                self._forward_backward(static_x, static_mask)
                self._allreduce()
                self._clip_and_update()
            self._restore(snap)                # the warm-up steps must not count as training steps
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph_fb, graph_opt = torch.cuda.CUDAGraph(), None
        if self.world == 1:
            with torch.cuda.graph(graph_fb):
                self._forward_backward(static_x, static_mask)
                self._clip_and_update()
        elif self.graph_allreduce:
            # RCCL's watchdog thread polls events while this thread captures: "thread_local" keeps its calls from invalidating
            # the capture (the default mode forbids CUDA/HIP calls from ANY thread of the process during capture)
            with torch.cuda.graph(graph_fb, capture_error_mode="thread_local"):
                self._forward_backward(static_x, static_mask)
                self._allreduce()
                self._clip_and_update()
        else:
            with torch.cuda.graph(graph_fb, capture_error_mode="thread_local"):
                self._forward_backward(static_x, static_mask)
            graph_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph_opt, pool=graph_fb.pool(), capture_error_mode="thread_local"):
                self._clip_and_update()
        return {"fb": graph_fb, "opt": graph_opt, "x": static_x, "mask": static_mask, "out": self.out}


def init_distributed(force_cpu=False):
    """One process per GPU; RCCL ('nccl' on ROCm) over xGMI. Returns (rank, world, device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not force_cpu and torch.cuda.is_available():
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    else:
        dev = torch.device("cpu")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = os.environ.get("KVAE_DIST_BACKEND", "nccl" if dev.type == "cuda" else "gloo")   # "nccl" == RCCL
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, dev


def train_one_epoch(trainer, loader, device, epoch=None):
    """Reference-shaped epoch loop (train.py:23-76): returns mean loss / elbo_kf / elbo_vae_total."""
    model = trainer.model
    model.train()
    if model.config.scheduled_beta and epoch is not None:
        trainer.set_beta(model.scheduler.get_beta(epoch))
    sums, n = None, 0
    for batch in loader:
        x = batch["images"].float().to(device, non_blocking=True)
        out = trainer.step(x)
        vals = torch.stack([out["loss"], out["elbo_kf"], out["elbo_vae_total"]])
        sums = vals.clone() if sums is None else sums + vals
        n += 1
    sums = (sums / max(n, 1)).tolist() if sums is not None else [0.0, 0.0, 0.0]
    return {"loss": sums[0], "elbo_kf": sums[1], "elbo_vae_total": sums[2]}


def end_of_epoch_schedules(trainer, scheduler, epoch, decay_steps=20, tau_decay_start_epoch=1):
    """What the reference's main loop does after train_one_epoch (train.py:268-274 there): every `decay_steps` epochs
    one step of the LR scheduler, and for switching dynamics the Gumbel-softmax temperature
    tau <- max(tau_min, tau * tau_decay_rate) every `tau_decay_steps` epochs from `tau_decay_start_epoch` on
    (the reference starts it after the two pre-training phases: max(1, pretrain_vae_epochs + warmup_epochs + 1), train.py:244).
    Both land in device scalars (the tensor lr of Adam, the tau scalar of the dynamics), so a step that has
    already been captured into a hipGraph follows them.  Returns (lr, tau) as floats for logging."""
    model = trainer.model
    cfg = model.config
    if scheduler is not None and epoch % decay_steps == 0:
        scheduler.step()
    dyn = model.kalman_filter.dyn_params
    tau = None
    if cfg.dynamics_model.lower() == "switching":
        if epoch % cfg.tau_decay_steps == 0 and epoch >= tau_decay_start_epoch \
                and (epoch - tau_decay_start_epoch) % cfg.tau_decay_steps == 0:
            dyn.tau = max(cfg.tau_min, dyn.tau * cfg.tau_decay_rate)
        tau = dyn.tau
    return float(trainer.opt.param_groups[0]["lr"]), tau
