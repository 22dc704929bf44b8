"""DNN magnitude-mask model of BASELINE config C0 -- stock PyTorch by design ("CPU plumbing config
only, no HIP"; reference: src/model/dnn.py:7-142, src/model/ema.py:4-39).

Same constructor arguments (incl. the reference's quirks: the YAML key ``n_layers`` is NOT the ctor's ``n_layer``, so 4 blocks
are always built; ``nfft`` stays 512; ``dnn_ema`` and ``n_fft`` are required keys), same forward
``[B, C, F, T, 2] -> [B, C, F, T, 2]`` and the same state_dict keys (``model.{n}.model.{0,1}.*``, ``context.*``,
``ema_{in,out}.ema{0,1}``), so the reference's checkpoints load.  The one change: the exponential moving average is the
closed form of the reference's Python time loop (a lower-triangular Toeplitz product), not 500 sequential tiny ops.
"""
import torch
from torch import nn


class ExponentialMovingAverage(nn.Module):
    """outputs_t = (1 - alpha) * outputs_{t-1} + alpha * inputs_t, outputs_0 = alpha * inputs_0 over dim 1 of [B, T, C]
    (src/model/ema.py:21-39) == sum_{s<=t} alpha (1-alpha)^(t-s) inputs_s."""

    def __init__(self, alpha, *args, **kwargs):
        super().__init__()
        self.alpha = alpha
        self.register_buffer("ema0", torch.ones(1) * alpha)
        self.register_buffer("ema1", torch.ones(1) * (1 - alpha))

    def forward(self, inputs):
        assert len(inputs.shape) == 3
        steps = inputs.shape[1]
        lag = torch.arange(steps, device=inputs.device)
        lag = lag[:, None] - lag[None, :]                                    # t - s
        kernel = torch.where(lag >= 0, self.ema0 * self.ema1 ** lag.clamp(min=0), torch.zeros((), device=inputs.device))
        return torch.matmul(kernel.to(inputs.dtype), inputs)


_ACTIVATIONS = {"linear": None, "leaky-relu": lambda: nn.LeakyReLU(negative_slope=0.1), "relu": nn.ReLU,
                "sigmoid": nn.Sigmoid, "tanh": nn.Tanh}


class NeuralNetwork(nn.Module):
    """Linear + BatchNorm1d (+ activation + Dropout unless last) on [rows, features] (src/model/dnn.py:7-63)."""

    def __init__(self, first=False, last=False, nfft=512, hidden_layer=1024, bias=True, activation="leaky-relu", drop_out=0,
                 *args, **kwargs):
        super().__init__()
        nfeature = int(nfft // 2 + 1)
        n_in = nfeature if first else hidden_layer
        n_out = hidden_layer if (first or not last) else nfeature
        layers = [nn.Linear(n_in, n_out, bias=bias), nn.BatchNorm1d(n_out)]
        if not last:
            if activation not in _ACTIVATIONS:
                raise ValueError(f"There is no implmentation for {activation}")
            if _ACTIVATIONS[activation] is not None:
                layers.append(_ACTIVATIONS[activation]())
            layers.append(nn.Dropout(p=drop_out))
        self.model = nn.ModuleList(layers)

    def forward(self, x):
        for layer in self.model:
            x = layer(x)
        return x


class DeepNeuralNetwork(nn.Module):
    def __init__(self, n_layer=4, dnn_method="mask", *args, **kwargs):
        super().__init__()
        self.model = nn.ModuleList([NeuralNetwork(*args, first=(n == 0), last=(n != 0 and n == n_layer - 1), **kwargs)
                                    for n in range(n_layer)])
        self.dnn_method = dnn_method
        self.ema = bool(kwargs["dnn_ema"])                      # required key, like the reference (src/model/dnn.py:86)
        if self.ema:
            n_feature = kwargs["n_fft"] // 2 + 1
            self.context = nn.Linear(n_feature, n_feature, bias=True)
            self.ema_in = ExponentialMovingAverage(alpha=0.1)
            self.ema_out = ExponentialMovingAverage(alpha=0.85)

    def forward(self, mix):
        batch, n_channel, n_feature, n_frame, _ = mix.shape
        x = torch.sqrt(mix[..., 0] ** 2 + mix[..., 1] ** 2)
        # (the reference squeezes when n_channel == 1, which breaks at batch 1, src/model/dnn.py:100-101; the reshape is
        #  identical for every batch > 1 and also works for 1)
        x = x.reshape(batch * n_channel, n_feature, n_frame).transpose(1, 2)
        if self.ema:
            x = self.ema_in(self.context(x))
        x = x.reshape(batch * n_channel * n_frame, n_feature)
        for layer in self.model:
            x = layer(x)
        x = x.reshape(batch * n_channel, n_frame, n_feature)
        if self.ema:
            x = self.ema_out(x)
        x = x.transpose(1, 2).reshape(batch, n_channel, n_feature, n_frame)
        if self.dnn_method == "mask":
            return mix * x.unsqueeze(-1)
        if self.dnn_method == "reconstruct":   # magnitudes carry no phase: the reference's exp(i*angle(x)) is sign(x)
            return torch.stack([x, torch.zeros_like(x)], dim=-1)
        return x
