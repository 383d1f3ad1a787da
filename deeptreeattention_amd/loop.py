"""Minimal Trainer-shaped loop around the fused step drivers (BASELINE.json configs[0] plumbing).

The reference trains through PyTorch-Lightning: `TreeModel.training_step / validation_step` (src/main.py:71-94) driven by
`Trainer.fit`, an Adam optimizer and `ReduceLROnPlateau(mode="min", factor=0.75, patience=8, threshold=1e-4 rel,
min_lr=1e-7, eps=1e-8)` monitoring `val_loss` (src/main.py:135-149; per level in src/models/multi_stage.py:258-275).
Lightning itself is out of scope; what a user of the hot path needs from it is small and lives here:

  * `PlateauScheduler`      - the same plateau rule, driving the `lr` of a fused trainer (which is not a
                              torch.optim.Optimizer, so torch's scheduler cannot attach to it);
  * `SyntheticTreeDataset`  - the batch structure of the reference's TreeDataset (src/data.py:284-310:
                              `(individual, {"HSI": ...}, label)`), synthetic, resident on the device;
  * `fit`                   - epochs of `training_step` / `validation_step` + the scheduler, as Trainer.fit runs them;
  * `fit_multistage`        - the same for the reference's MultiStage module (train.py:75-100): every level of a batch in one
                              launch chain, per-level validation loaders and plateau schedulers.
No arithmetic happens here; every step is the HIP path behind `engine.FusedTrainer` and friends.
"""
import torch


class PlateauScheduler:
    """torch.optim.lr_scheduler.ReduceLROnPlateau with the reference's settings, for objects exposing a float `lr`
    attribute (FusedTrainer, EnsembleTrainer, MetadataTrainer).  `step(metric)` once per validation epoch."""

    def __init__(self, trainer, mode="min", factor=0.75, patience=8, threshold=1e-4, threshold_mode="rel", cooldown=0,
                 min_lr=1e-7, eps=1e-8):
        if factor >= 1.0:
            raise ValueError("Factor should be < 1.0.")
        if mode not in ("min", "max") or threshold_mode not in ("rel", "abs"):
            raise ValueError("unknown mode")
        self.trainer = trainer
        self.mode, self.factor, self.patience = mode, float(factor), int(patience)
        self.threshold, self.threshold_mode = float(threshold), threshold_mode
        self.cooldown, self.min_lr, self.eps = int(cooldown), float(min_lr), float(eps)
        self.best = float("inf") if mode == "min" else -float("inf")
        self.num_bad_epochs = 0
        self.cooldown_counter = 0
        self.last_epoch = 0

    def _is_better(self, a):
        if self.mode == "min":
            ref = self.best * (1.0 - self.threshold) if self.threshold_mode == "rel" else self.best - self.threshold
            return a < ref
        ref = self.best * (1.0 + self.threshold) if self.threshold_mode == "rel" else self.best + self.threshold
        return a > ref

    def step(self, metric):
        current = float(metric)          # a 0-d device tensor is read here: one host sync per EPOCH, as in Lightning
        self.last_epoch += 1
        if self._is_better(current):
            self.best = current
            self.num_bad_epochs = 0
        else:
            self.num_bad_epochs += 1
        if self.cooldown_counter > 0:
            self.cooldown_counter -= 1
            self.num_bad_epochs = 0
        if self.num_bad_epochs > self.patience:
            old = float(self.trainer.lr)
            new = max(old * self.factor, self.min_lr)
            if old - new > self.eps:
                self.trainer.lr = new
            self.cooldown_counter = self.cooldown
            self.num_bad_epochs = 0
        return float(self.trainer.lr)


class SyntheticTreeDataset:
    """Synthetic stand-in for the reference's TreeDataset (src/data.py:284-310): item = (individual, inputs, label) with
    `inputs["HSI"]` a (bands, H, W) float32 patch in [0, 1) (real crops are min-max scaled per pixel, src/utils.py:49) or,
    with `years > 0`, a list of per-year patches as TreeDataset yields them for learned_ensemble (`missing` = fraction of
    year crops zero-filled the way TreeDataset fills years without imagery).  Patches are generated once on the device
    from a seeded generator; `loader(batch_size)` yields collated batches the way the default collate would:
    (list of individual ids, {"HSI": (B, bands, H, W) tensor or list of them [, "site": (B,) int64]}, (B,) int64)."""

    def __init__(self, n, bands, classes, size=11, years=0, sites=0, missing=0.0, seed=0, device="cuda"):
        g = torch.Generator(device=device)
        g.manual_seed(int(seed))
        self.n, self.years, self.sites = int(n), int(years), int(sites)
        shape = (self.n, bands, size, size)
        if years:
            self.hsi = [torch.rand(shape, device=device, generator=g) for _ in range(years)]
            if missing > 0:
                # whole batches share the years they have in practice (one flight campaign per site and year); here
                # individuals are dropped independently, which only makes all-zero year BATCHES rarer
                for t in self.hsi[1:]:
                    gone = torch.rand(self.n, device=device, generator=g) < missing
                    t[gone] = 0
        else:
            self.hsi = torch.rand(shape, device=device, generator=g)
        self.labels = torch.randint(0, classes, (self.n,), device=device, generator=g)
        self.site = torch.randint(0, sites, (self.n,), device=device, generator=g) if sites else None
        self.individuals = ["tree_{}".format(i) for i in range(self.n)]

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        inputs = {"HSI": [t[i] for t in self.hsi] if self.years else self.hsi[i]}
        if self.site is not None:
            inputs["site"] = self.site[i]
        return self.individuals[i], inputs, self.labels[i]

    def loader(self, batch_size, shuffle=False, seed=0, drop_last=False):
        idx = torch.arange(self.n, device=self.labels.device)
        if shuffle:
            g = torch.Generator(device=self.labels.device)
            g.manual_seed(int(seed))
            idx = torch.randperm(self.n, device=self.labels.device, generator=g)
        for lo in range(0, self.n, batch_size):
            sel = idx[lo:lo + batch_size]
            if drop_last and sel.numel() < batch_size:
                break
            hsi = [t[sel] for t in self.hsi] if self.years else self.hsi[sel]
            inputs = {"HSI": hsi}
            if self.site is not None:
                inputs["site"] = self.site[sel]
            yield [self.individuals[int(i)] for i in sel.tolist()], inputs, self.labels[sel]


def fit(trainer, train_data, val_data=None, epochs=1, batch_size=64, scheduler=None, shuffle=True, log=None):
    """Epoch loop as Lightning's Trainer.fit runs the reference's LightningModules: `training_step(batch, i)` on every
    training batch, then `validation_step(batch, i)` on every validation batch, `val_loss` = mean of the batch losses
    (what `self.log("val_loss", loss)` aggregates, src/main.py:90), then the plateau scheduler.  `trainer` is a
    FusedTrainer / MetadataTrainer (or anything with the two step methods).  Losses stay on the device inside an epoch
    (one host read per epoch).  Returns a list of {"epoch", "train_loss", "val_loss", "lr"} records."""
    history = []
    for epoch in range(int(epochs)):
        tl = [trainer.training_step(b, i) for i, b in enumerate(train_data.loader(batch_size, shuffle, seed=epoch))]
        rec = {"epoch": epoch, "train_loss": float(torch.stack([t.reshape(()) for t in tl]).mean()), "val_loss": None}
        if val_data is not None:
            model = getattr(trainer, "model", None)
            was_training = bool(model.training) if model is not None else False
            if model is not None:
                model.eval()                 # Lightning's validation loop: running BatchNorm statistics, no updates
            try:
                vl = [trainer.validation_step(b, i) for i, b in enumerate(val_data.loader(batch_size))]
            finally:
                if model is not None and was_training:
                    model.train()
            rec["val_loss"] = float(torch.stack([t.reshape(()).float() for t in vl]).mean())
            if scheduler is not None:
                scheduler.step(rec["val_loss"])
        rec["lr"] = float(trainer.lr)
        history.append(rec)
        if log:
            log(rec)
    return history


def fit_multistage(trainer, train_data, val_data=None, epochs=1, batch_size=128, schedulers=None, shuffle=True, log=None):
    """The epoch loop `train.py:75-100` runs for the reference's MultiStage module: `train_dataloader()` returns one loader
    per level (multi_stage.py:212-229), Lightning zips them into a list-of-batches and calls `training_step` once per
    optimizer -- here ALL levels of a batch are one launch chain (`MultiStageTrainer.training_step_all`; a level whose
    loader ran out is cycled, as Lightning's "max_size_cycle" mode does) --, then `validation_step(batch, i, level)` over every
    level's own validation loader (multi_stage.py:231-246, :290-304), `val_loss/dataloader_idx_{level}` = mean of its batch
    losses, and one plateau scheduler per level monitoring it (multi_stage.py:258-275).
    trainer: engine.MultiStageTrainer; train_data / val_data: one SyntheticTreeDataset(years=...) (or anything with
    `.loader(batch_size, shuffle, seed)`) per level; schedulers: one PlateauScheduler per level (on `trainer.levels[l]`) or
    None.  Returns a list of {"epoch", "train_loss": [...], "val_loss": [...], "lr": [...]} records."""
    import itertools
    nl = len(trainer.levels)
    if len(train_data) != nl or (val_data is not None and len(val_data) != nl):
        raise ValueError("one dataset per level ({})".format(nl))
    history = []
    for epoch in range(int(epochs)):
        loaders = [list(d.loader(batch_size, shuffle, seed=epoch)) for d in train_data]
        steps = max(len(b) for b in loaders)
        sums = [[] for _ in range(nl)]
        for i in range(steps):
            batch = [b[i % len(b)] for b in loaders]               # shorter loaders cycle ("max_size_cycle")
            for l, loss in enumerate(trainer.training_step_all(batch, i)):
                sums[l].append(loss.reshape(()))
        rec = {"epoch": epoch, "train_loss": [float(torch.stack(s).mean()) for s in sums], "val_loss": None}
        if val_data is not None:
            models = [t.model for t in trainer.levels]
            was = [bool(m.training) for m in models]
            for m in models:
                m.eval()                     # Lightning's validation loop: running BatchNorm statistics, no updates
            try:
                vl = []
                for l, d in enumerate(val_data):
                    out = [trainer.validation_step(b, i, l)["val_loss"].reshape(()).float() for i, b in enumerate(d.loader(batch_size))]
                    vl.append(float(torch.stack(out).mean()))
            finally:
                for m, w in zip(models, was):
                    if w:
                        m.train()
            rec["val_loss"] = vl
            if schedulers is not None:
                for l, sch in enumerate(schedulers):
                    if sch is not None:
                        sch.step(vl[l])
        rec["lr"] = [float(t.lr) for t in trainer.levels]
        history.append(rec)
        if log:
            log(rec)
    return history
