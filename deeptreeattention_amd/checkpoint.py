"""Checkpoint interoperability with the reference.

The module classes here keep the reference's state_dict names, so `load_state_dict` / `Hang2020.load_from_backbone`
already exchange weights with `src/models/Hang2020.py` and `src/models/year.py`.  What is left is the key prefixing of
the reference's LightningModules:
  * MultiStage (src/models/multi_stage.py:21, :41, :66): `models.{level}.model.year_models.{year}.<spectral_network key>`
    inside the Lightning checkpoint's "state_dict" (`MultiStage.load_from_checkpoint`, `tests/test_multi_stage.py:17-19`,
    `notebooks/embeddings.py:13-14`), next to per-level `loss_weight_{level}` buffers;
  * TreeModel (src/main.py:33-69): `model.<Hang2020 key>`.
These helpers move tensors between such checkpoints and the modules of this package, both directions."""
import torch


def _state_dict_of(obj):
    if isinstance(obj, (str, bytes)) or hasattr(obj, "__fspath__"):
        obj = torch.load(obj, map_location="cpu")
    if isinstance(obj, dict) and "state_dict" in obj and isinstance(obj["state_dict"], dict):
        obj = obj["state_dict"]           # a Lightning checkpoint
    return obj


def load_multistage(checkpoint, levels, strict=True):
    """Load a reference MultiStage checkpoint (path, Lightning checkpoint dict or bare state_dict) into `levels`, a
    list of deeptreeattention_amd.year.learned_ensemble, one per hierarchical level.  Returns the per-level class
    weight tensors found in the checkpoint ({level: tensor}), which the reference stores as `loss_weight_{level}`."""
    sd = _state_dict_of(checkpoint)
    for i, ens in enumerate(levels):
        prefix = "models.{}.model.".format(i)
        sub = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
        if not sub and strict:
            raise KeyError("checkpoint holds no '{}*' entries".format(prefix))
        ens.load_state_dict(sub, strict=strict)
    return {int(k.rsplit("_", 1)[1]): v for k, v in sd.items() if k.startswith("loss_weight_")}


def multistage_state_dict(levels, loss_weights=None):
    """The inverse: a state_dict with the reference MultiStage's key names, ready for `torch.save` or for
    `MultiStage.load_state_dict` on the reference side."""
    out = {}
    for i, ens in enumerate(levels):
        for k, v in ens.state_dict().items():
            out["models.{}.model.{}".format(i, k)] = v.detach().cpu()
    for i, w in (loss_weights or {}).items():
        out["loss_weight_{}".format(i)] = torch.as_tensor(w).detach().cpu()
    return out


def load_treemodel(checkpoint, model, strict=True):
    """Load a reference TreeModel checkpoint (`model.<key>` entries, src/main.py:33) into a network of this package."""
    sd = _state_dict_of(checkpoint)
    sub = {k[len("model."):]: v for k, v in sd.items() if k.startswith("model.")}
    if not sub and strict:
        raise KeyError("checkpoint holds no 'model.*' entries")
    return model.load_state_dict(sub, strict=strict)
