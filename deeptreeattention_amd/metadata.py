"""Site-metadata late fusion around the HIP-backed Hang2020 (drop-in for the model classes of the reference's
`src/models/metadata.py`; parameter names and shapes are the reference's, so state_dicts interoperate).

Only the sensor branch carries real work (>99.9 % of the FLOPs) and it is the HIP path.  The site branch - a 16-wide
embedding, BatchNorm1d, Dropout(0.7) and one Linear - and the 2*classes -> classes fusion layer are <0.2 MFLOP per
sample.  At MODULE level (this file: `model(images, site)` under autograd) they are stock torch modules
that hold the parameters under the reference's names; inside the fused step -- `engine.MetadataTrainer`, the reference's
MetadataModel.training_step (metadata.py:52-63: unweighted cross-entropy of `model(images, site)`) -- the same layers
run natively by default (`native_head=True`: csrc/meta.hip, dta_meta_head_forward / _loss / _backward, 9 launches, checked
against these torch modules' autograd and the reference's golden), and `native_head=False` keeps the torch graph.  The
reference's MetadataModel LightningModule shell itself is out of scope."""
import torch
from torch import nn

from .Hang2020 import Hang2020

_SITE_WIDTH = 16          # reference metadata.py:12
_SITE_DROPOUT = 0.7       # reference metadata.py:15


class metadata(nn.Module):
    """Site index -> (B, classes) scores: embedding -> BN -> dropout -> linear -> ReLU (reference metadata.py:9-24)."""

    def __init__(self, sites, classes):
        super().__init__()
        self.embedding = nn.Embedding(sites, _SITE_WIDTH)
        self.batch_norm = nn.BatchNorm1d(_SITE_WIDTH)
        self.mlp = nn.Linear(_SITE_WIDTH, classes)
        self.dropout = nn.Dropout(_SITE_DROPOUT)

    def forward(self, x):
        # embedding(x) written as one_hot(x) @ weight: the same values (one weight row plus zeros), but the backward is a
        # small deterministic GEMM instead of torch's dense embedding backward (49 us per 1024 x 16 step on MI355X, and
        # float atomics): the metadata train step is run-to-run reproducible like the rest
        onehot = nn.functional.one_hot(x, self.embedding.num_embeddings).to(self.embedding.weight.dtype)
        return torch.relu(self.mlp(self.dropout(self.batch_norm(onehot @ self.embedding.weight))))


class metadata_sensor_fusion(nn.Module):
    """ReLU(Linear(cat[site scores, Hang2020 scores])) (reference metadata.py:26-44)."""

    def __init__(self, bands, sites, classes, precision=None):
        super().__init__()
        self.metadata_model = metadata(sites, classes)
        self.sensor_model = Hang2020(bands, classes, precision)      # the HIP path
        self.fc1 = nn.Linear(2 * classes, classes)

    def forward(self, images, metadata):
        joined = torch.cat((self.metadata_model(metadata), self.sensor_model(images)), dim=1)
        return torch.relu(self.fc1(joined))
