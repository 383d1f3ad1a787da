"""Minimal Trainer-shaped loop around the fused step drivers (BASELINE.json configs[0] plumbing).

The reference trains through PyTorch-Lightning: `TreeModel.training_step / validation_step` (src/main.py:71-94) driven by
`Trainer.fit`, an Adam optimizer and `ReduceLROnPlateau(mode="min", factor=0.75, patience=8, threshold=1e-4 rel,
min_lr=1e-7, eps=1e-8)` monitoring `val_loss` (src/main.py:135-149; per level in src/models/multi_stage.py:258-275).
Lightning itself is out of scope; what a user of the hot path needs from it is small and lives here:

  * `PlateauScheduler`      - the same plateau rule, driving the `lr` of a fused trainer (which is not a
                              torch.optim.Optimizer, so torch's scheduler cannot attach to it);
  * `SyntheticTreeDataset`  - the batch structure of the reference's TreeDataset (src/data.py:284-310:
                              `(individual, {"HSI": ...}, label)`), synthetic, resident on the device;
  * `fit`                   - epochs of `training_step` / `validation_step` + the scheduler, as Trainer.fit runs them.
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
