"""Drop-in for the model classes of the reference's `src/models/metadata.py` (site-embedding late fusion).

`metadata_sensor_fusion.sensor_model` is the HIP-backed Hang2020; the 16-wide site MLP and the 2*classes->classes
fusion layer (<0.2 MFLOP per sample, SURVEY.md 8 a13: "fusion MLP is tiny, may stay torch") are stock torch modules
with the reference's names, so state_dicts interoperate.  (The reference's MetadataModel LightningModule shell is
out of scope; its training_step is `F.cross_entropy(model(images, metadata), y)`, metadata.py:52-63.)"""
import torch
from torch import nn
from torch.nn import functional as F

from .Hang2020 import Hang2020


class metadata(nn.Module):
    """reference metadata.py:9-24"""

    def __init__(self, sites, classes):
        super().__init__()
        self.embedding = nn.Embedding(sites, 16)
        self.batch_norm = nn.BatchNorm1d(16)
        self.mlp = nn.Linear(in_features=16, out_features=classes)
        self.dropout = nn.Dropout(p=0.7)

    def forward(self, x):
        x = self.embedding(x)
        x = self.batch_norm(x)
        x = self.dropout(x)
        x = self.mlp(x)
        return F.relu(x)


class metadata_sensor_fusion(nn.Module):
    """reference metadata.py:26-44"""

    def __init__(self, bands, sites, classes, precision=None):
        super().__init__()
        self.metadata_model = metadata(sites, classes)
        self.sensor_model = Hang2020(bands, classes, precision)
        self.fc1 = nn.Linear(in_features=classes * 2, out_features=classes)

    def forward(self, images, metadata):
        metadata_softmax = self.metadata_model(metadata)
        sensor_softmax = self.sensor_model(images)
        concat_features = torch.cat([metadata_softmax, sensor_softmax], dim=1)
        return F.relu(self.fc1(concat_features))
