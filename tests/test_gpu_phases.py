"""GPU tier (-m gpu): the captured `Trainer` against what the REFERENCE recorded - the one training step of trainstep_*.npz
(kvae/train/train.py:44-58 there) and the three training phases of phases_*.npz (train.py:142-207, 246-260), both through the
path bench.py times (hipGraph, LGSSM chain on the side stream with its early backward, fused loss head, flat clip + Adam) and
through the eager path; plus what the Trainer's step schedule must NOT leak into: a plain training-mode use of the model,
a second Trainer on the same model, resume through the flat optimizer, a last partial batch."""
import pytest
import torch

import parity_cases
from golden_util import load, rel_err, sub

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _fixture_model(g, kind):
    from kvae.model.model import KVAE
    from kvae.utils.config import KVAEConfig
    model = KVAE(KVAEConfig(dynamics_model=kind, num_modes=3, scheduled_beta=False))
    model.load_state_dict(sub(g, "sd."), strict=True)
    model.to(DEV).train()
    model.beta = float(g["beta"])
    return model


class _Feed:
    """Static device buffers for the frames and every injected draw, so that a captured step reads new values each time."""

    def __init__(self, **first):
        self.buf = {k: v.float().to(DEV).clone() for k, v in first.items() if v is not None}

    def set(self, **vals):
        for k, v in vals.items():
            if v is not None:
                self.buf[k].copy_(v.float().to(DEV))

    def noise(self):
        return {k: self.buf.get(k) for k in ("eps_a", "eps_z", "gumbel")}


@pytest.mark.parametrize("use_graph", [True, False])
@pytest.mark.parametrize("name,kind", [("trainstep_lstm_K3", "lstm"), ("trainstep_switch_K3", "switching")])
def test_trainer_reproduces_the_recorded_reference_step(name, kind, use_graph):
    """`Trainer(use_graph=True)` - side stream, early LGSSM backward, KVAE_LOSS_HEAD=2, kvae_clip_adam: what bench.py times -
    for ONE step on the reference's recorded step: loss / elbo_kf <= 1e-4, grad norm <= 1e-3, post-Adam parameters <= 1e-3."""
    from kvae import noise
    from kvae.train.train import Trainer
    g = load(name)
    model = _fixture_model(g, kind)
    tr = Trainer(model, lr=float(g["lr"]), grad_clip_norm=float(g["clip"]), use_graph=use_graph)
    assert (tr.lgssm_stream is not None) is use_graph and tr.early_kf_backward is use_graph and tr._flat_step
    feed = _Feed(x=g["frames"], eps_a=g["eps_a"], eps_z=g["eps_z"], gumbel=g.get("gumbel"))
    with noise.inject(**feed.noise()):
        out = tr.step(feed.buf["x"])
    torch.cuda.synchronize()
    assert rel_err(out["loss"].cpu(), g["loss"]) < 1e-4
    assert rel_err(out["elbo_kf"].cpu(), g["elbo_kf"]) < 1e-4
    assert rel_err(out["elbo_vae_total"].cpu(), g["elbo_vae"]) < 1e-4
    assert rel_err(out["grad_norm"].cpu(), g["grad_norm"]) < 1e-3
    names = [k for k, _ in model.named_parameters()]
    # the fixture holds the gradients AFTER clip_grad_norm_ scaled them in place; the flat buffer is left unscaled (kvae_clip_adam
    # applies the factor on the fly)
    scale = min(1.0, float(g["clip"]) / (float(out["grad_norm"]) + 1e-6))
    for k, gv in zip(names, tr.grad_views):
        ref = float(g["gradnorm." + k])
        assert abs(float(gv.norm()) * scale - ref) <= 2e-3 * ref + 1e-6, k
    sd = model.state_dict()
    checked = 0
    for k in names:
        if "after." + k in g:
            # Adam's first update is lr*g/(|g|+1e-8): entries whose gradient is O(1e-8) are rounding-sensitive, hence 1e-3
            assert rel_err(sd[k].cpu(), g["after." + k]) < 1e-3, k
            checked += 1
        ref = float(g["afternorm." + k])
        assert abs(float((sd[k].cpu() - g["sd." + k]).norm()) - ref) <= 2e-2 * ref + 1e-7, k
    assert checked >= 10
    assert [float(s) for s in tr._seg_steps.cpu()] == [1.0] * len(names)   # ONE step, not the capture warm-up's four


@pytest.mark.parametrize("use_graph", [True, False])
@pytest.mark.parametrize("name,kind", [("phases_lstm_K3", "lstm"), ("phases_switch_K3", "switching")])
def test_training_phases_gpu(name, kind, use_graph):
    """vae -> warmup -> all, two steps each, against the reference's own set_training_phase + train_one_epoch: epoch means,
    which parameters moved (frozen ones stay BIT-identical: no moment, step or value update), the parameters after each phase
    and Adam's per-parameter step counts.  The captured trainer re-captures once per phase."""
    from kvae import noise
    from kvae.train.train import Trainer
    g = load(name)
    model = _fixture_model(g, kind)
    tr = Trainer(model, lr=float(g["lr"]), grad_clip_norm=float(g["clip"]), use_graph=use_graph)
    feed = _Feed(x=g["frames0"], eps_a=g["vae.eps_a0"], eps_z=g["vae.eps_z0"], gumbel=g.get("vae.gumbel0"))
    names = [k for k, _ in model.named_parameters()]
    captures = []   # the trainer's capture count after every step

    def set_phase(phase):
        tr.set_training_phase(phase)
        assert [int(p.requires_grad) for p in model.parameters()] == g[f"{phase}.trainable"].tolist()
        assert tr.kf_weight == float(g[f"{phase}.kf_weight"])

    def run_step(phase, i, kf_weight):
        feed.set(x=g[f"frames{i}"], eps_a=g[f"{phase}.eps_a{i}"], eps_z=g[f"{phase}.eps_z{i}"], gumbel=g.get(f"{phase}.gumbel{i}"))
        with noise.inject(**feed.noise()):
            out = tr.step(feed.buf["x"])
        captures.append(tr.captures)
        return {k: float(out[k]) for k in ("loss", "elbo_kf", "elbo_vae_total")}

    parity_cases.check_phases(g, set_phase, run_step, lambda: {k: p.detach().cpu() for k, p in model.named_parameters()},
                              lambda: tr._seg_steps.cpu().tolist(), value_tol=1e-4)
    assert captures == ([1, 1, 2, 2, 3, 3] if use_graph else [0] * 6)   # one capture per phase, replayed for its second step
    # the optimizer's state reads like the reference's: per-parameter steps, no moments on what never had a gradient
    st = tr.opt.state_dict()["state"]
    for i, k in enumerate(names):
        assert float(st[i]["step"]) == float(g["adam_steps"][i]), k


def test_vae_phase_launches_no_lgssm_backward():
    """"vae" phase: kf_weight = 0 and every LGSSM parameter frozen - the chain runs forward only (for the logged elbo_kf), and
    the frozen alpha-network's weight-gradient products are skipped in "warmup"."""
    from kvae import _native
    from kvae.train.synthetic import bouncing_ball
    from kvae.train.train import Trainer
    from test_gpu_training_features import _model
    x = bouncing_ball(8, 10, 5).float().to(DEV)
    tr = Trainer(_model("lstm"), use_graph=False)
    seen = {}
    for phase in ("vae", "warmup", "all"):
        tr.set_training_phase(phase)
        tr.step(x)
        _native.profile_start()
        tr.step(x)
        seen[phase] = set(_native.profile_stop())
    assert "smooth_fwd" in seen["vae"] and "elbo" in seen["vae"]
    assert not {"smooth_bwd", "lstm_bwd"} & seen["vae"]
    assert {"smooth_bwd", "lstm_bwd"} <= seen["warmup"] and "rnn_wgrad" not in seen["warmup"]
    assert {"smooth_bwd", "lstm_bwd", "rnn_wgrad"} <= seen["all"]


@pytest.mark.parametrize("kind", ["lstm", "switching"])
def test_model_outside_its_trainer_is_the_plain_model(kind):
    """After a Trainer has been built on (and has stepped) a model, a training-mode forward + a custom loss on the smoothed
    means + backward OUTSIDE the trainer must give the encoder the LGSSM's gradient as on an untouched model: the trainer's
    schedule (side stream, early LGSSM backward) lives only inside its own step."""
    from kvae import noise
    from kvae.train.synthetic import bouncing_ball
    from kvae.train.train import Trainer
    from test_gpu_training_features import _model, _noise
    B, T = 6, 9
    x = bouncing_ball(B, T, 3).float().to(DEV)
    nz = _noise(B, T, seed=2)

    def custom_grads(model):
        for p in model.parameters():
            p.grad = None
        with noise.inject(**nz):
            out = model(x)
        (out["mus_smooth"].square().sum() + out["x_logits"].mean()).backward()
        return {k: p.grad.detach().clone().cpu() for k, p in model.named_parameters() if p.grad is not None}

    plain = custom_grads(_model(kind, seed=4))
    model = _model(kind, seed=4)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    tr = Trainer(model, use_graph=True)
    with noise.inject(**nz):
        tr.step(x)
    torch.cuda.synchronize()
    assert model.lgssm_stream is None and model.early_kf_backward is False and model.kf_value_only is False
    model.load_state_dict(sd0)   # undo the optimizer step: same weights as `plain`
    after = custom_grads(model)
    assert set(after) == set(plain)
    assert float(plain["encoder.fc_mu.weight"].abs().max()) > 0
    for k in plain:
        assert rel_err(after[k], plain[k]) < 1e-5, k
    with noise.inject(**nz):   # and compute_loss without a backward leaves no gradient behind on the dynamics parameters
        for p in model.parameters():
            p.grad = None
        model.compute_loss(x, model(x), with_metrics=False)
    assert all(p.grad is None for p in model.kalman_filter.parameters())


def test_second_trainer_supersedes_the_first():
    from kvae.train.synthetic import bouncing_ball
    from kvae.train.train import Trainer
    from test_gpu_training_features import _model
    x = bouncing_ball(4, 8, 1).float().to(DEV)
    model = _model("lstm")
    first = Trainer(model, use_graph=True)
    first.step(x)
    second = Trainer(model, use_graph=False)
    with pytest.raises(RuntimeError, match="superseded"):
        first.step(x)
    before = torch.cat([p.detach().flatten() for p in model.parameters()]).clone()
    second.step(x)
    torch.cuda.synchronize()
    assert float((torch.cat([p.detach().flatten() for p in model.parameters()]) - before).abs().max()) > 0


@pytest.mark.parametrize("same_trainer", [False, True])
def test_resume_through_the_flat_optimizer(tmp_path, same_trainer):
    """Two captured steps, save (reference-format payload), load - into a fresh Trainer, or back into the same, already
    captured one after a third step has moved it on - and compare the next step with the uninterrupted run.  The file also
    must not turn the optimizer into a non-capturable one (a reference-written file says fused None, capturable False)."""
    from kvae import noise
    from kvae.train.checkpoint import Checkpointer, load_checkpoint
    from kvae.train.synthetic import bouncing_ball
    from kvae.train.train import Trainer
    from test_gpu_training_features import _model, _noise
    B, T = 6, 9
    x = bouncing_ball(B, T, 3).float().to(DEV)
    nz = _noise(B, T, seed=21)
    flat = lambda m: torch.cat([p.detach().flatten() for p in m.parameters()]).cpu()

    model = _model("lstm", seed=8)
    tr = Trainer(model, lr=3e-3, use_graph=True)
    tr.set_training_phase("warmup")   # a frozen alpha-network: its optimizer state must stay empty across save / load
    with noise.inject(**nz):
        tr.step(x), tr.step(x)
        path = tmp_path / "ck.pt"
        Checkpointer(tmp_path).save_checkpoint(path, model, tr.opt, 2, 0.0, 0.0)
        tr.step(x)
        torch.cuda.synchronize()
        want, want_steps = flat(model), tr._seg_steps.cpu().tolist()
        payload = torch.load(path, weights_only=True)
        frozen = [i for i, p in enumerate(model.parameters()) if not p.requires_grad]
        assert frozen and all(i not in payload["optimizer_state"]["state"] for i in frozen)
        assert all(isinstance(v, float) or not torch.is_tensor(v) for gr in payload["optimizer_state"]["param_groups"]
                   for k, v in gr.items() if k != "params")
        if same_trainer:
            tr2, model2 = tr, model
        else:
            model2 = _model("lstm", seed=99)
            tr2 = Trainer(model2, lr=1e-1, use_graph=True)
            tr2.set_training_phase("warmup")
        load_checkpoint(path, model2, tr2.opt, map_location=DEV)
        gr = tr2.opt.param_groups[0]
        assert gr["capturable"] is True and gr["fused"] is True and torch.is_tensor(gr["lr"]) and abs(float(gr["lr"]) - 3e-3) < 1e-9
        tr2.step(x)
        torch.cuda.synchronize()
    assert tr2._seg_steps.cpu().tolist() == want_steps
    assert float((flat(model2) - want).abs().max()) < 1e-6


def test_partial_last_batch_gets_its_own_captured_step():
    """A batch of another size (the last partial batch of an epoch) must not fail on the static buffers of the captured step:
    it is captured on its own, and going back to the full size replays the first capture."""
    from kvae import noise
    from kvae.train.synthetic import bouncing_ball
    from kvae.train.train import Trainer
    from test_gpu_training_features import _model, _noise
    xs = {8: bouncing_ball(8, 10, 5).float().to(DEV), 3: bouncing_ball(3, 10, 6).float().to(DEV)}
    nzs = {b: _noise(b, 10, seed=b) for b in xs}

    def run(use_graph):
        model = _model("lstm", seed=2)
        tr = Trainer(model, lr=2e-3, use_graph=use_graph)
        graphs = []
        for b in (8, 3, 8, 3):
            with noise.inject(**nzs[b]):
                tr.step(xs[b])
            graphs.append(tr.captures)
        torch.cuda.synchronize()
        return torch.cat([p.detach().flatten() for p in model.parameters()]).cpu(), graphs

    pg, graphs = run(True)
    pe, _ = run(False)
    assert graphs == [1, 2, 2, 2]   # one capture per batch shape; going back to a shape replays its graph
    assert float((pg - pe).abs().max()) < 2e-4
