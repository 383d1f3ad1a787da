"""ORACLE / CPU BASELINE (test infrastructure, never shipped as the product path).

PyTorch-eager functional restatement of the reference's Hang2020 train step, written over a flat
{state_dict name: tensor} dict instead of the reference's nn.Module classes.  It exists for one purpose the
NumPy oracle cannot serve: bench.py's `cpu_baseline` leg needs the reference's own execution model (stock
torch ops dispatched to oneDNN on the host cores, autograd backward, torch.optim.Adam) timed on the GPU box,
where /root/reference does not exist.  tests/test_oracle_golden.py pins it to the golden vectors produced by
the reference itself.  Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import it.

Reference lines restated: src/models/Hang2020.py:24-31 (conv_module), :105-124 (spatial_attention),
:149-168 (spectral_attention), :190-204/:226-240 (networks), :251-263 (Hang2020), src/main.py:71-80 (step),
:135-137 (Adam).
"""
import torch
import torch.nn.functional as F

_SPATIAL_POOL = {32: 4, 64: 2, 128: 1}


def _conv_block(p, pre, x, pool, training):
    y = F.conv2d(x, p[pre + "conv_layer.weight"], p[pre + "conv_layer.bias"], padding=1)
    y = F.batch_norm(y, p[pre + "bn1.running_mean"], p[pre + "bn1.running_var"], p[pre + "bn1.weight"],
                     p[pre + "bn1.bias"], training, 0.1, 1e-5)
    if training:
        p[pre + "bn1.num_batches_tracked"] += 1
    y = F.relu(y)
    return F.max_pool2d(y, 2) if pool else y


def _spectral_gate(p, pre, z):
    v = z.mean(dim=(2, 3)).unsqueeze(-1)
    k = p[pre + "attention_conv1.weight"].shape[-1]
    h = F.relu(F.conv1d(v, p[pre + "attention_conv1.weight"], p[pre + "attention_conv1.bias"], padding=k // 2))
    g = torch.sigmoid(F.conv1d(h, p[pre + "attention_conv2.weight"], p[pre + "attention_conv2.bias"], padding=k // 2))
    a = z * g.unsqueeze(-1)
    return a, a.mean(dim=(2, 3))


def _spatial_gate(p, pre, z):
    m = F.relu(F.conv2d(z, p[pre + "channel_pool.weight"], p[pre + "channel_pool.bias"]))
    k = p[pre + "attention_conv1.weight"].shape[-1]
    t = F.relu(F.conv2d(m, p[pre + "attention_conv1.weight"], p[pre + "attention_conv1.bias"], padding=k // 2))
    s = torch.sigmoid(F.conv2d(t, p[pre + "attention_conv2.weight"], p[pre + "attention_conv2.bias"], padding=k // 2))
    a = z * s
    return a, torch.flatten(F.max_pool2d(a, _SPATIAL_POOL[z.shape[1]]), 1)


def subnet(p, pre, kind, x, training):
    gate = _spectral_gate if kind == "spectral" else _spatial_gate
    scores = []
    u = x
    for L in (1, 2, 3):
        z = _conv_block(p, f"{pre}conv{L}.", u, L > 1, training)
        u, f = gate(p, f"{pre}attention_{L}.", z)
        scores.append(F.linear(f, p[f"{pre}classifier{L}.fc1.weight"], p[f"{pre}classifier{L}.fc1.bias"]))
    return scores


def hang2020(p, x, training=True):
    spec = subnet(p, "spectral_network.", "spectral", x, training)[-1]
    spat = subnet(p, "spatial_network.", "spatial", x, training)[-1]
    w = torch.sigmoid(p["alpha"])
    return spec * w + spat * (1 - w)


def to_tensors(params_np):
    """NumPy parameter dict -> torch tensors (trainable ones require grad)."""
    out = {}
    for k, v in params_np.items():
        t = torch.tensor(v)
        if t.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            t.requires_grad_(True)
        out[k] = t
    return out


class TrainStep:
    """forward + weighted CE + backward + Adam: the reference's TreeModel step on stock torch ops."""

    def __init__(self, params, lr, loss_weight=None):
        self.p = params
        self.w = loss_weight
        self.opt = torch.optim.Adam([t for t in params.values() if t.requires_grad], lr=lr)

    def __call__(self, x, y):
        self.opt.zero_grad(set_to_none=True)
        logits = hang2020(self.p, x, True)
        loss = F.cross_entropy(logits, y, weight=self.w)
        loss.backward()
        self.opt.step()
        return logits.detach(), loss.detach()


def learned_ensemble(p, images, training=True, pre=""):
    """src/models/year.py:24-33: one spectral_network per year; a year whose whole batch tensor sums to zero is
    skipped (:27); the kept years' last-head scores are averaged (:30, :33)."""
    scores = []
    for i, x in enumerate(images):
        if x.sum() == 0:
            continue
        scores.append(subnet(p, f"{pre}year_models.{i}.", "spectral", x, training)[-1])
    return torch.stack(scores, dim=1).mean(dim=1)


class EnsembleTrainStep:
    """One level of the reference's MultiStage loop on stock torch ops: src/models/multi_stage.py:277-288
    (training_step: weighted CE of the ensemble's scores) and :258-275 (one Adam per level).  Parameters of a
    skipped year keep grad None, so torch's Adam leaves them, their moments and their step counts untouched."""

    def __init__(self, params, lr, loss_weight=None):
        self.p = params
        self.w = loss_weight
        self.opt = torch.optim.Adam([t for t in params.values() if t.requires_grad], lr=lr)

    def __call__(self, images, y):
        self.opt.zero_grad(set_to_none=True)
        scores = learned_ensemble(self.p, images, True)
        loss = F.cross_entropy(scores, y, weight=self.w)
        loss.backward()
        self.opt.step()
        return scores.detach(), loss.detach()


def metadata_sensor_fusion(p, images, site, training=True, dropout_p=0.7, pre=""):
    """src/models/metadata.py:9-44 over a flat parameter dict: site branch = Embedding(sites, 16) -> BatchNorm1d ->
    Dropout(0.7) -> Linear(16, classes) -> ReLU (:17-24); sensor branch = Hang2020 (:31, :39); fusion =
    ReLU(Linear(cat([site, sensor], 1))) (:40-42)."""
    e = F.embedding(site, p[pre + "metadata_model.embedding.weight"])
    e = F.batch_norm(e, p[pre + "metadata_model.batch_norm.running_mean"], p[pre + "metadata_model.batch_norm.running_var"],
                     p[pre + "metadata_model.batch_norm.weight"], p[pre + "metadata_model.batch_norm.bias"],
                     training, 0.1, 1e-5)
    e = F.dropout(e, dropout_p, training)
    meta = F.relu(F.linear(e, p[pre + "metadata_model.mlp.weight"], p[pre + "metadata_model.mlp.bias"]))
    sensor = hang2020({k[len(pre + "sensor_model."):]: v for k, v in p.items() if k.startswith(pre + "sensor_model.")},
                      images, training)
    return F.relu(F.linear(torch.cat([meta, sensor], dim=1), p[pre + "fc1.weight"], p[pre + "fc1.bias"]))
