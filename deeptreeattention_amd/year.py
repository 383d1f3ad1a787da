"""MI355X-native drop-in for the reference's `src/models/year.py` (year ensemble).

One `spectral_network` per year; a year is skipped only when its WHOLE batch tensor sums to zero
(reference year.py:27); the kept years' last-head scores are averaged (:30, :33).  Each sub-network runs through
the HIP library (only head 3 is evaluated, as the reference discards heads 1-2); the 3-way stack/mean over
(B, classes) scores is torch plumbing."""
import torch
from torch import nn

from . import Hang2020


class learned_ensemble(nn.Module):
    def __init__(self, years, classes, config):
        super().__init__()
        self.year_models = nn.ModuleList()
        self.years = years
        for _ in range(years):
            if config["pretrain_state_dict"]:
                base_model = Hang2020.load_from_backbone(state_dict=config["pretrain_state_dict"], classes=classes,
                                                         bands=config["bands"])
            else:
                base_model = Hang2020.spectral_network(bands=config["bands"], classes=classes)
            self.year_models.append(base_model)

    def forward(self, images):
        # same test as the reference (year.py:27: a year is skipped iff its whole batch tensor sums to zero), but all
        # years' sums travel to the host in ONE transfer instead of one blocking comparison per year
        keep = (torch.stack([x.sum() for x in images]) != 0).tolist()
        year_scores = [self.year_models[index]._run(x, 4)[0] for index, x in enumerate(images) if keep[index]]
        return torch.stack(year_scores, axis=1).mean(axis=1)
