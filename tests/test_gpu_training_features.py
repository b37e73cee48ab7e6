"""GPU tier (-m gpu): the callers either side of the LGSSM path on the real device - schedules that must reach a
captured hipGraph (learning rate, Gumbel temperature), the device-side data path, checkpoints written by the
reference, and the reference-shaped loop body with an explicit mask."""
import numpy as np
import pytest
import torch

from golden_util import GOLDEN, load, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model(kind, seed=0, **cfg_kw):
    from kvae.model.model import KVAE
    from kvae.utils.config import KVAEConfig
    torch.manual_seed(seed)
    m = KVAE(KVAEConfig(dynamics_model=kind, num_modes=3, **cfg_kw))
    with torch.no_grad():
        m.kalman_filter.dyn_params.A.add_(0.05 * torch.randn_like(m.kalman_filter.dyn_params.A))
        if hasattr(m.kalman_filter.dyn_params, "head_w"):
            m.kalman_filter.dyn_params.head_w.bias.zero_()
    m.beta = 1.0
    return m.to(DEV).train()


def _noise(B, T, n=4, K=3, seed=1):
    g = torch.Generator().manual_seed(seed)
    return dict(eps_a=torch.randn(B * T, 2, generator=g).to(DEV), eps_z=torch.randn(B, T, n, generator=g).to(DEV),
                gumbel=(-torch.empty(B, T, K).exponential_(generator=g).log()).to(DEV))


@pytest.mark.parametrize("kind", ["switching", "lstm"])
def test_lr_and_tau_schedules_reach_the_captured_graph(kind):
    """Two 'epochs' of two steps with the reference's end-of-epoch schedules (train.py:268-274: ExponentialLR step,
    tau <- max(tau_min, tau * rate)) between them: the hipGraph-replayed trainer must land on the same parameters
    as the eager one (both consume the same injected noise), and must differ from a run WITHOUT the schedules -
    i.e. the decayed values really are read by the replayed kernels, not baked in at capture."""
    from kvae import noise
    from kvae.train.synthetic import bouncing_ball
    from kvae.train.train import Trainer, end_of_epoch_schedules
    B, T = 8, 10
    x = bouncing_ball(B, T, 5).float().to(DEV)
    nz = _noise(B, T)

    def run(use_graph, schedules):
        model = _model(kind)
        cfg = model.config
        cfg.tau_decay_steps, cfg.tau_decay_rate, cfg.tau_min = 1, 0.5, 0.05
        tr = Trainer(model, lr=5e-3, use_graph=use_graph, overlap_lgssm=False)
        sched = torch.optim.lr_scheduler.ExponentialLR(tr.opt, gamma=0.25)
        taus = []
        with noise.inject(**nz):
            for epoch in (1, 2):
                for _ in range(2):
                    tr.step(x)
                if schedules:
                    lr, tau = end_of_epoch_schedules(tr, sched, epoch, decay_steps=1, tau_decay_start_epoch=1)
                    taus.append(tau)
        torch.cuda.synchronize()
        flat = torch.cat([p.detach().flatten() for p in model.parameters()]).cpu()
        return flat, float(tr.opt.param_groups[0]["lr"]), taus

    eager, lr_e, taus_e = run(False, True)
    graph, lr_g, taus_g = run(True, True)
    frozen, _, _ = run(True, False)
    assert abs(lr_e - 5e-3 * 0.25 ** 2) < 1e-9 and abs(lr_g - lr_e) < 1e-12
    assert taus_e == taus_g
    if kind == "switching":
        assert taus_g[1] == pytest.approx(0.5 * taus_g[0]) and taus_g[1] < taus_g[0]          # tau really decayed
    # Adam normalises the step size, so graph == eager up to fp32 reassociation of a few 1e-3-sized updates
    assert float((graph - eager).abs().max()) < 2e-4, "captured step did not follow the schedules"
    assert float((graph - frozen).abs().max()) > 2e-3, "schedules had no effect on the replayed graph"


def test_tau_assignment_is_read_by_a_captured_regime_chain():
    """The reference assigns `dyn_params.tau = ...` (train.py:274); a RegimeChain launch captured BEFORE the
    assignment must use the new temperature when replayed."""
    from kvae.kalman.lgssm_ops import RegimeChain
    from kvae.kalman.switch_dyn_param import StickyRegimePrior, SwitchingDynamicsParameter
    K, B, T = 3, 4, 6
    g = torch.Generator().manual_seed(0)
    logits, init = torch.randn(B, T, K, K, generator=g).to(DEV), torch.randn(B, K, generator=g).to(DEV)
    gum = (-torch.empty(B, T, K).exponential_(generator=g).log()).to(DEV)
    dyn = SwitchingDynamicsParameter(torch.eye(4).repeat(K, 1, 1), torch.zeros(K, 4, 4), torch.zeros(K, 2, 4),
                                     prior=StickyRegimePrior(K, 0.8)).to(DEV)
    dyn.tau = 1.0
    P = dyn.prior.transition_matrix.to(DEV)
    tau_t = dyn.tau_scalar(torch.device(DEV, torch.cuda.current_device()))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.no_grad():
        RegimeChain.apply(logits, init, gum, P, tau_t, False)
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph), torch.no_grad():
        y, lq, lp = RegimeChain.apply(logits, init, gum, P, tau_t, False)
    graph.replay()
    y1 = y.clone()
    dyn.tau = 0.3
    graph.replay()
    y2 = y.clone()
    with torch.no_grad():
        want1, _, _ = RegimeChain.apply(logits, init, gum, P, 1.0, False)    # float tau: baked into the launch
        want2, _, _ = RegimeChain.apply(logits, init, gum, P, 0.3, False)
    assert rel_err(y1.cpu(), want1.cpu()) < 1e-6 and rel_err(y2.cpu(), want2.cpu()) < 1e-6
    assert float((y1 - y2).abs().max()) > 1e-2


@pytest.mark.parametrize("resident", [True, False])
def test_device_batches_on_gpu_match_items(tmp_path, resident):
    """DeviceBatches in resident mode (uint8 data set in HBM, gather + normalise on the GPU) and streaming mode (pinned
    host memory, uint8 over PCIe on a copy stream one batch ahead) both deliver exactly ds[i]."""
    from kvae.dataloader.pymunk_dataset import DeviceBatches, PymunkNPZDataset
    rng = np.random.default_rng(0)
    p = tmp_path / "d.npz"
    np.savez_compressed(p, images=rng.integers(0, 256, size=(13, 6, 32, 32), dtype=np.uint8))
    ds = PymunkNPZDataset.from_npz(str(p), seq_len=6, state_key=None)
    loader = DeviceBatches(ds, 4, DEV, shuffle=True, seed=7, resident=resident, drop_last=False)
    assert loader.resident is resident
    order = torch.randperm(13, generator=torch.Generator().manual_seed(7))
    seen = 0
    for bi, batch in enumerate(loader):
        x = batch["images"]
        assert x.is_cuda and x.dtype == torch.float32 and x.shape[1:] == (6, 1, 32, 32)
        for j in range(x.shape[0]):
            assert torch.equal(x[j].cpu(), ds[int(order[bi * 4 + j])]["images"])
        seen += x.shape[0]
    assert seen == 13
    assert len(list(DeviceBatches(ds, 4, DEV, shuffle=False, resident=resident, drop_last=True))) == 3


@pytest.mark.parametrize("kind", ["lstm", "switching"])
def test_reference_checkpoint_reproduces_reference_loss(kind):
    """Load the file the REFERENCE's Checkpointer wrote (weights_only=True) and run the forward the reference ran
    after saving it (tests/golden/make_goldens_r2.py): loss, ELBO parts and smoothed means must match."""
    from kvae import noise
    from kvae.model.model import KVAE
    from kvae.train.checkpoint import load_checkpoint
    from kvae.utils.config import KVAEConfig
    g = load(f"ref_checkpoint_{kind}")
    cfg = KVAEConfig(dynamics_model=kind, num_modes=3, encoder_channels=g["encoder_channels"].tolist(),
                     decoder_channels=g["decoder_channels"].tolist(), dynamics_hidden_dim=int(g["dynamics_hidden_dim"]))
    with pytest.warns(UserWarning):
        model = KVAE(cfg)
    load_checkpoint(GOLDEN / f"ref_checkpoint_{kind}.pt", model)
    model.to(DEV).train()
    model.beta = float(g["beta"])
    x = g["frames"].float().to(DEV)
    with torch.no_grad(), noise.inject(eps_a=g["eps_a"], eps_z=g["eps_z"], gumbel=g.get("gumbel")):
        out = model(x, mask=torch.ones(x.shape[:2], device=DEV))
        losses = model.compute_loss(x, out, mask=torch.ones(x.shape[:2], device=DEV), with_metrics=False)
    assert rel_err(out["a_samples"].cpu(), g["a_samples"]) < 2e-5
    assert rel_err(out["mus_smooth"].cpu(), g["mus_smooth"]) < 1e-4
    for k, gk in (("loss", "loss"), ("elbo_kf", "elbo_kf"), ("elbo_vae_total", "elbo_vae")):
        assert rel_err(losses[k].cpu(), g[gk]) < 1e-4, k


@pytest.mark.parametrize("kind", ["lstm", "switching"])
def test_reference_shaped_loop_body(kind):
    """The body of the reference's train_one_epoch verbatim (train.py:32-62: reset_state, explicit mask of ones,
    zero_grad(set_to_none), model(x, mask=mask), compute_loss, backward, clip_grad_norm_, optimizer.step, three
    float(loss) reads) on the drop-in classes, 3 steps: runs without a host-side mask check, losses finite and the
    first one equal to the Trainer's step on the same noise."""
    from kvae import noise
    from kvae.train.synthetic import bouncing_ball
    from kvae.train.train import Trainer
    B, T = 6, 12
    x = bouncing_ball(B, T, 9).float().to(DEV)
    nz = _noise(B, T, seed=4)
    model = _model(kind, seed=3)
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=0.0)
    losses = []
    with noise.inject(**nz):
        for _ in range(3):
            model.kalman_filter.dyn_params.reset_state()
            mask = torch.ones(B, T, device=DEV)
            optimizer.zero_grad(set_to_none=True)
            outputs = model(x, mask=mask)
            out = model.compute_loss(x, outputs, kf_weight=1.0, vae_weight=1.0, mask=mask)
            out["loss"].backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
            optimizer.step()
            losses.append((float(out["loss"].detach()), float(out["elbo_kf"].detach()), float(out["elbo_vae_total"].detach())))
    assert all(np.isfinite(v) for row in losses for v in row)
    assert isinstance(out["active_units"], int)
    ref_model = _model(kind, seed=3)
    tr = Trainer(ref_model, lr=1e-3, use_graph=False)
    with noise.inject(**nz):
        first = tr.step(x)
    assert abs(float(first["loss"]) - losses[0][0]) < 1e-4 * abs(losses[0][0])


def test_flat_clip_adam_equals_torch_adam(monkeypatch):
    """kvae_clip_adam (clip_grad_norm_ + Adam on flat buffers, two launches) against torch.linalg.vector_norm + torch's fused Adam
    on the same gradients: three eager steps with a learning-rate change in between, weight decay on; parameters, moments and the
    reported gradient norm must agree to rounding, and the optimizer's state_dict must read like an ordinary Adam's."""
    from kvae import noise
    from kvae.train.synthetic import bouncing_ball
    from kvae.train.train import Trainer
    B, T = 6, 9
    x = bouncing_ball(B, T, 3).float().to(DEV)
    nz = _noise(B, T, seed=11)

    def run(flat):
        monkeypatch.setenv("KVAE_FLAT_ADAM", "1" if flat else "0")
        model = _model("lstm", seed=5)
        tr = Trainer(model, lr=3e-3, weight_decay=1e-3, grad_clip_norm=0.5, use_graph=False)
        assert tr._flat_step is flat
        norms = []
        with noise.inject(**nz):
            for i in range(3):
                out = tr.step(x)
                norms.append(float(out["grad_norm"]))
                if i == 0:
                    tr.set_lr(1e-3)
        sd = tr.opt.state_dict()
        flatp = torch.cat([p.detach().flatten() for p in model.parameters()]).cpu()
        m = torch.cat([sd["state"][i]["exp_avg"].flatten() for i in range(len(tr.params))]).cpu()
        v = torch.cat([sd["state"][i]["exp_avg_sq"].flatten() for i in range(len(tr.params))]).cpu()
        return flatp, m, v, norms, float(sd["state"][0]["step"])

    pf, mf, vf, nf, sf = run(True)
    pt, mt, vt, nt, st = run(False)
    assert sf == st == 3.0
    assert max(abs(a - b) / b for a, b in zip(nf, nt)) < 1e-5 and nt[0] > 0.5          # the clip was active
    assert float((pf - pt).abs().max()) < 2e-6
    assert rel_err(mf, mt) < 1e-5 and rel_err(vf, vt) < 5e-5   # (the second moment squares the clip scale: twice its rounding)


@pytest.mark.parametrize("kind", ["lstm", "switching"])
def test_early_lgssm_backward_gives_the_same_step(kind, monkeypatch):
    """KVAE.early_kf_backward (the LGSSM branch differentiated on the side stream right behind its forward, its gradient w.r.t.
    the encodings handed to loss.backward() at the join) against the ordinary single backward: same loss, same flat gradient
    after one captured step, same parameters after three - with an explicit mask, both dynamics models."""
    from kvae import noise
    from kvae.train.synthetic import bouncing_ball
    from kvae.train.train import Trainer
    B, T = 6, 9
    x = bouncing_ball(B, T, 3).float().to(DEV)
    mask = (torch.rand(B, T, generator=torch.Generator().manual_seed(3)) > 0.2).float().to(DEV)
    nz = _noise(B, T, seed=12)

    def run(early):
        monkeypatch.setenv("KVAE_EARLY_KF_BWD", "1" if early else "0")
        model = _model(kind, seed=6)
        tr = Trainer(model, lr=3e-3, grad_clip_norm=10.0, use_graph=True)
        assert tr.lgssm_stream is not None and tr.early_kf_backward is early and model.lgssm_stream is None   # the schedule lives in the trainer
        with noise.inject(**nz):
            out = tr.step(x, mask)
            loss1, kf1, grad1 = float(out["loss"]), float(out["elbo_kf"]), tr.flat_grad.detach().clone().cpu()
            for _ in range(2):
                tr.step(x, mask)
        torch.cuda.synchronize()
        return loss1, kf1, grad1, torch.cat([p.detach().flatten() for p in model.parameters()]).cpu()

    le, ke, ge, pe = run(True)
    ll, kl, gl, pl = run(False)
    assert abs(le - ll) <= 1e-6 * abs(ll) and abs(ke - kl) <= 1e-6 * abs(kl)
    assert float(gl.abs().max()) > 0 and rel_err(ge, gl) < 1e-5
    assert float((pe - pl).abs().max()) < 1e-4   # three Adam steps at lr 3e-3: rounding-level gradient differences move entries with near-zero gradient
