"""Task-embedding CNN of MMAML (reference: networks/conv_embedding_model.py:15-188), module-level
parity target X1 of SURVEY.md §8a.

Constructor surface and state_dict keys (`conv.conv{i}`, `conv.bn{i}`, `linear`, `_embeddings.{i}`)
follow the reference for the configuration the repo uses (MMAMLShapeNet1D.py:63-81: convolutional,
batch-norm, avg-pool after conv, no RNN).
"""
from collections import OrderedDict

import torch

from mlhot.ops import AggFunction, BatchNormReluFunction, Conv2dFunction, LinearFunction, SpatialMeanFunction


class ConvEmbeddingModel(torch.nn.Module):
    def __init__(self, input_size, output_size, embedding_dims, hidden_size=128, num_layers=1, convolutional=False,
                 num_conv=4, num_channels=32, num_channels_max=256, rnn_aggregation=False, linear_before_rnn=False,
                 embedding_pooling="max", batch_norm=True, avgpool_after_conv=True, num_sample_embedding=0,
                 sample_embedding_file="embedding.hdf5", img_size=(1, 28, 28), verbose=False):
        super().__init__()
        if not convolutional or rnn_aggregation or not avgpool_after_conv or not batch_norm:
            raise NotImplementedError("mlhot implements the convolutional / batch-norm / avg-pool-after-conv / "
                                      "no-RNN configuration the reference instantiates (MMAMLShapeNet1D.py:63-81)")
        self._input_size, self._output_size, self._hidden_size = input_size, output_size, hidden_size
        self._embedding_dims, self._num_conv, self._img_size = embedding_dims, num_conv, img_size
        self._embedding_pooling = embedding_pooling
        self._device = "cpu"
        chans = [img_size[0]] + [min(num_channels_max, num_channels * 2 ** i) for i in range(num_conv)]
        chans = [min(num_channels_max, c) for c in chans]
        layers = OrderedDict()
        for i in range(num_conv):
            layers[f"conv{i + 1}"] = torch.nn.Conv2d(chans[i], chans[i + 1], (3, 3), stride=2, padding=1)
            layers[f"bn{i + 1}"] = torch.nn.BatchNorm2d(chans[i + 1], momentum=0.001)
            layers[f"relu{i + 1}"] = torch.nn.ReLU(inplace=True)
        self.conv = torch.nn.Sequential(layers)
        self.rnn = None
        self.linear = torch.nn.Linear(chans[-1], hidden_size)
        self.relu_after_linear = torch.nn.ReLU(inplace=True)
        self._embeddings = torch.nn.ModuleList([torch.nn.Linear(hidden_size, d) for d in embedding_dims])

    def forward(self, x, params=None, return_task_embedding=False):
        """x: the shots of ONE task [n, C, H, W] -> list of 4 embedding vectors [1, dim]
        (conv_embedding_model.py:99-184; `params` may override the module's own parameters)."""
        if params is None:
            params = OrderedDict(self.named_parameters())
        for i in range(1, self._num_conv + 1):
            x = Conv2dFunction.apply(x, params[f"conv.conv{i}.weight"], params[f"conv.conv{i}.bias"], 2, 1, False)
            bn = getattr(self.conv, f"bn{i}")
            # F.batch_norm(training=True) in the reference: batch statistics, default momentum 0.1, in-place running update
            x = BatchNormReluFunction.apply(x, params[f"conv.bn{i}.weight"], params[f"conv.bn{i}.bias"], bn.running_mean,
                                            bn.running_var, 0.1, 1e-5)
        x = SpatialMeanFunction.apply(x)
        hid = LinearFunction.apply(x, params["linear.weight"], params["linear.bias"], "relu")
        if self._embedding_pooling not in ("avg", "max"):
            raise NotImplementedError
        pooled, _ = AggFunction.apply("mean" if self._embedding_pooling == "avg" else "max", hid[None], None)   # over the shot axis
        out = [LinearFunction.apply(pooled, params[f"_embeddings.{j}.weight"], params[f"_embeddings.{j}.bias"], "none")
               for j in range(len(self._embeddings))]
        return (out, pooled) if return_task_embedding else out

    def to(self, device, **kwargs):
        self._device = device
        return super().to(device, **kwargs)
