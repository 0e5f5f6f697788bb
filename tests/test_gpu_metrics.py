"""Device-side validation / evaluation metrics (csrc/metrics.hip through the C-ABI) against the reference's golden
values (G9) and the numpy oracle. Tolerances: the reference computes in fp32 two-pass, the kernels in fp64 one-pass;
1e-4 relative on scalars, 2e-5 absolute on correlations, 2e-4 on FEV ratios (fp32 cancellation in the reference)."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import metrics_oracle as MO
from oracle import v1t_oracle as O
from oracle import weights as W
from tests.helpers import assert_close, build_native_model

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "g9_metrics.npz")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _ds(d, tier="test", hashed=False):
    return SimpleNamespace(dataset=SimpleNamespace(tier=tier, hashed=hashed, neuron_ids=d["neuron_ids"].copy(), mouse_id="A"))


def test_compute_metrics_vs_reference_golden(dev):
    from v1t_amd.metrics import compute_metrics

    g, d = np.load(GOLD), MO.make_metric_data()
    r = compute_metrics(y_true=torch.from_numpy(d["targets"]).to(dev), y_pred=torch.from_numpy(d["predictions"]).to(dev))
    for k in ("metrics/msse", "metrics/poisson_loss", "metrics/single_trial_correlation"):
        assert abs(float(r[k]) - float(g[f"g9/{k}"])) <= 1e-4 * abs(float(g[f"g9/{k}"])), (k, float(r[k]), float(g[f"g9/{k}"]))


def test_metrics_class_vs_reference_golden(dev):
    """Metrics mirror: unordered rows / neuron ids in, the reference's neuron-ordered per-neuron vectors out."""
    from v1t_amd.metrics import Metrics

    g, d = np.load(GOLD), MO.make_metric_data()
    res = {"predictions": torch.from_numpy(d["predictions"]).to(dev), "targets": torch.from_numpy(d["targets"]).to(dev),
           "image_ids": torch.from_numpy(d["image_ids"]), "trial_ids": torch.from_numpy(d["trial_ids"])}
    m = Metrics(_ds(d), res)
    assert_close("stc", m.single_trial_correlation(per_neuron=True), g["g9/single_trial_correlation"], 0, 2e-5)
    assert_close("cta", m.correlation_to_average(per_neuron=True), g["g9/correlation_to_average"], 0, 2e-5)
    kept = m.feve(per_neuron=True)
    assert kept.shape == g["g9/feve_kept"].shape
    assert_close("feve", kept, g["g9/feve_kept"], 2e-4, 2e-4)
    assert abs(m.feve() - g["g9/feve_kept"].mean()) < 1e-4
    # train / validation tiers have no repeats: the reference returns None
    mv = Metrics(_ds(d, tier="validation"), res)
    assert mv.correlation_to_average() is None and mv.feve() is None
    with pytest.raises(RuntimeError):
        Metrics(_ds(d), {**res, "predictions": res["predictions"].cpu()})


def test_streaming_ragged_micro_batches_match_one_shot(dev):
    """Folding ragged micro-batches (1, 7, 64, rest) gives the same moments as one launch over all trials."""
    from v1t_amd.metrics import StreamingMetrics

    d = MO.make_metric_data(seed=3, images=15, repeats=7, neurons=1000)
    p, y = torch.from_numpy(d["predictions"]).to(dev), torch.from_numpy(d["targets"]).to(dev)
    one = StreamingMetrics(1000, dev, image_groups=True, max_images=15)
    one.update(p, y, image_ids=d["image_ids"])
    parts = StreamingMetrics(1000, dev, image_groups=True, max_images=15)
    i = 0
    for n in (1, 7, 64, 10 ** 6):
        parts.update(p[i:i + n], y[i:i + n], image_ids=d["image_ids"][i:i + n])
        i += n
    assert parts.count == one.count == 105
    assert_close("corr", parts.correlation(), one.correlation(), 0, 1e-6)
    for a, b in zip(parts.repeat_statistics(), one.repeat_statistics()):
        assert_close("repeat", a, b, 1e-6, 1e-6)
    ot, op, oi = MO.order(d["targets"], d["predictions"], d["image_ids"], d["trial_ids"], np.arange(1000))
    assert_close("corr.oracle", one.correlation(), MO.correlation(d["predictions"], d["targets"], axis=0), 0, 2e-5)
    assert_close("cta.oracle", one.repeat_statistics()[0], MO.correlation_to_average(ot, op, oi), 0, 2e-5)
    assert abs(float(one.msse()) - float(MO.msse(d["targets"], d["predictions"]))) <= 1e-4 * float(one.msse())
    with pytest.raises(RuntimeError):
        StreamingMetrics(10, torch.device("cpu"))


class _Dataset(SimpleNamespace):
    def __len__(self):
        return 12


class _Loader:
    """Minimal DataLoader stand-in: iterable of batches + .dataset with the attributes the reference reads."""

    def __init__(self, batches, dataset):
        self._b, self.dataset = batches, dataset

    def __iter__(self):
        return iter(self._b)

    def __len__(self):
        return len(self._b)


def test_validate_and_evaluate_loops_vs_oracle(dev):
    """validate (train.py:160-190) and evaluate (utils/utils.py:103-199) over two mice with ragged last batches: losses,
    msse, poisson loss, correlation and the challenge metrics against the oracle model + numpy metric oracle."""
    from v1t_amd.evaluate import evaluate, validate
    from v1t_amd.losses import PoissonLoss

    cfg = O.Config(num_blocks=1, emb_dim=64, mlp_dim=128, num_heads=4, mouse_ids=("A", "B"), num_neurons={"A": 96, "B": 50})
    sd = W.make_state_dict(cfg, 21)
    model, args = build_native_model(cfg, sd, dev)
    args.batch_size, args.micro_batch_size = 5, 3
    rng = np.random.default_rng(0)
    loaders, expect, ev = {}, {}, {}
    for mouse, n in cfg.num_neurons.items():
        full = W.make_batch(cfg, mouse, 12, 21)
        full["image_id"] = torch.from_numpy(np.repeat(np.arange(4), 3)[rng.permutation(12)])
        full["trial_id"] = torch.from_numpy(rng.permutation(12))
        batches = [{k: v[i:i + 5] for k, v in full.items()} for i in range(0, 12, 5)]  # 5, 5, 2
        dsn = _Dataset(tier="test", hashed=False, neuron_ids=rng.permutation(n) + 7, mouse_id=mouse)
        loaders[mouse] = _Loader(batches, dsn)
        with torch.no_grad():
            y = O.model_forward(cfg, sd, full["image"], mouse, full["behavior"], full["pupil_center"]).numpy()
        t_ = full["response"].numpy()
        reg = float(O.regularizer(cfg, sd, mouse))
        losses = []
        for b in batches:
            bs = b["image"].shape[0]
            for i in range(0, bs, 3):
                sl = slice(i, i + 3)
                yy = O.model_forward(cfg, sd, b["image"][sl], mouse, b["behavior"][sl], b["pupil_center"][sl])
                losses.append((float(O.poisson_loss(b["response"][sl], yy, 12.0, bs)), yy.shape[0] / bs * reg))
        # train.py:155 gathers (sums) micro-batches per batch, log_metrics averages over batches
        per_batch, k = [], 0
        for b in batches:
            nmb = -(-b["image"].shape[0] // 3)
            per_batch.append((sum(x[0] for x in losses[k:k + nmb]), sum(x[1] for x in losses[k:k + nmb])))
            k += nmb
        expect[mouse] = {"loss": np.mean([x[0] for x in per_batch]), "reg_loss": np.mean([x[1] for x in per_batch]), **MO.compute_metrics(t_, y)}
        ot, op, oi = MO.order(t_, y, full["image_id"].numpy(), full["trial_id"].numpy(), dsn.neuron_ids)
        ev[mouse] = (MO.correlation(op, ot, axis=0).mean(), MO.correlation_to_average(ot, op, oi).mean(), MO.feve(ot, op, oi))
    crit = PoissonLoss(args, loaders).to(dev)
    res = validate(args, loaders, model, crit)
    assert set(res) == {"loss", "reg_loss", "total_loss", "msse", "poisson_loss", "single_trial_correlation"}
    for key, ok in (("loss", "loss"), ("reg_loss", "reg_loss"), ("msse", "metrics/msse"), ("poisson_loss", "metrics/poisson_loss"),
                    ("single_trial_correlation", "metrics/single_trial_correlation")):
        want = np.mean([expect[m][ok] for m in cfg.mouse_ids])
        assert abs(res[key] - want) <= 2e-3 * abs(want) + 1e-5, (key, res[key], want)
    out = evaluate(args, loaders, model)
    assert abs(out["single_trial_correlation"] - np.mean([ev[m][0] for m in cfg.mouse_ids])) < 2e-3
    assert abs(out["correlation_to_average"] - np.mean([ev[m][1] for m in cfg.mouse_ids])) < 2e-3
    if all(len(ev[m][2]) for m in cfg.mouse_ids):
        want = np.mean([ev[m][2].mean() for m in cfg.mouse_ids])
        assert abs(out["feve"] - want) <= 5e-3 * max(1.0, abs(want))
