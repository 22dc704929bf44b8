"""CPU oracle for Demucs forward / loss / gradients (BASELINE config C3) -- TEST INFRASTRUCTURE ONLY.

Functional fp32 PyTorch-CPU restatement of the reference's time-domain Demucs: reference src/model/demucs.py:272-501 (Demucs:
constructor :362-428, valid_length :430-451, forward :453-490), :139-207 (DConv residual branches), :73-120 (BLSTM with its
overlapping chunks of max_steps), :210-269 (LocalState attention, nfreqs = 0), :52-71 (LayerScale), :17-50 (unfold,
center_trim).  Parameters live in one dict keyed by the reference's state_dict names (``encoder.4.3.layers.1.3.lstm.
weight_ih_l0_reverse`` ...).  It shares no code with the reference.  Only tests/ may import it.

Parity pinning
  * Everything except the resampler is pinned: tests/test_oracle_golden.py::test_demucs_oracle_matches_reference checks
    every encoder / decoder output, the separated sources, a loss and every parameter gradient against
    tests/golden/demucs_tiny.npz, which oracle/gen_golden_demucs.py produced from the IMPORTED reference with
    ``resample=False`` (two cases: one with LSTM chunking, T > max_steps).
  * ``resample=True`` calls julius.resample_frac (src/model/demucs.py:470,486), a third-party package (the reference's
    README.md:108 pins julius==0.2.7) that is absent from /root/reference and from this image, and no reference test holds an
    output of it.  `resample_frac` below restates julius 0.2.7's published algorithm (ResampleFrac: windowed-sinc kernels with
    zeros=24, rolloff=0.945, a squared-cosine window, kernels normalised to unit sum, replicate padding, one strided
    convolution per output phase).  **Parity unpinned for the resampler**: only its mathematical properties are tested
    (DC gain 1, a band-limited sine survives x2 then /2, output lengths).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


class DemucsConfig:
    """The constructor arguments that shape the network (src/model/demucs.py:273-309)."""

    def __init__(self, sources, audio_channels=2, channels=64, growth=2.0, depth=6, rewrite=True, lstm_layers=0, kernel_size=8,
                 stride=4, context=1, gelu=True, glu=True, norm_starts=4, norm_groups=4, dconv_mode=1, dconv_depth=2, dconv_comp=4,
                 dconv_attn=4, dconv_lstm=4, dconv_init=1e-4, normalize=True, resample=True, rescale=0.1, **_ignored):
        self.sources = list(sources)
        self.S = len(self.sources)
        self.audio_channels, self.channels, self.growth, self.depth = audio_channels, channels, growth, depth
        self.rewrite, self.lstm_layers, self.kernel_size, self.stride, self.context = rewrite, lstm_layers, kernel_size, stride, context
        self.gelu, self.glu, self.norm_starts, self.norm_groups = gelu, glu, norm_starts, norm_groups
        self.dconv_mode, self.dconv_depth, self.dconv_comp = dconv_mode, dconv_depth, dconv_comp
        self.dconv_attn, self.dconv_lstm, self.dconv_init = dconv_attn, dconv_lstm, dconv_init
        self.normalize, self.resample, self.rescale = normalize, resample, rescale

    def layer_channels(self):
        """[(in_channels, channels)] per encoder index (src/model/demucs.py:377-419)."""
        out, cin, ch = [], self.audio_channels, self.channels
        for _ in range(self.depth):
            out.append((cin, ch))
            cin, ch = ch, int(self.growth * ch)
        return out

    def valid_length(self, length):
        """src/model/demucs.py:430-451"""
        if self.resample:
            length *= 2
        for _ in range(self.depth):
            length = max(1, math.ceil((length - self.kernel_size) / self.stride) + 1)
        for _ in range(self.depth):
            length = (length - 1) * self.stride + self.kernel_size
        if self.resample:
            length = math.ceil(length / 2)
        return int(length)


# ---------------------------------------------------------------------------------------------------------------------------
# julius.resample_frac restated (julius 0.2.7, resample.py: class ResampleFrac) -- parity unpinned, see the module docstring
# ---------------------------------------------------------------------------------------------------------------------------
def resample_kernels(old_sr: int, new_sr: int, zeros: int = 24, rolloff: float = 0.945):
    """The `new_sr` interpolation kernels [new_sr, 2*width + old_sr] and `width`."""
    g = math.gcd(old_sr, new_sr)
    old_sr, new_sr = old_sr // g, new_sr // g
    sr = min(new_sr, old_sr) * rolloff
    width = math.ceil(zeros * old_sr / sr)
    idx = torch.arange(-width, width + old_sr, dtype=torch.float32)
    ks = []
    for i in range(new_sr):
        t = (-i / new_sr + idx / old_sr) * sr
        t = t.clamp(-zeros, zeros) * math.pi
        window = torch.cos(t / zeros / 2) ** 2
        k = torch.where(t == 0, torch.ones_like(t), torch.sin(t) / torch.where(t == 0, torch.ones_like(t), t)) * window
        ks.append(k / k.sum())
    return torch.stack(ks), width, old_sr, new_sr


def resample_frac(x, old_sr: int, new_sr: int):
    """x [..., T] at old_sr -> [..., floor(T * new_sr / old_sr)] at new_sr."""
    if old_sr == new_sr:
        return x
    kernels, width, old_sr, new_sr = resample_kernels(old_sr, new_sr)
    shape, length = x.shape[:-1], x.shape[-1]
    y = F.pad(x.reshape(-1, 1, length), (width, width + old_sr), mode="replicate")
    y = F.conv1d(y, kernels.to(x)[:, None], stride=old_sr)                    # [*, new_sr, frames]
    y = y.transpose(1, 2).reshape(*shape, -1)
    return y[..., :int(new_sr * length / old_sr)]


# ---------------------------------------------------------------------------------------------------------------------------
def unfold(a, kernel_size, stride):
    """[*, T] -> [*, ceil(T / stride), kernel_size] frames of the zero-padded input (src/model/demucs.py:17-32)."""
    length = a.shape[-1]
    n = math.ceil(length / stride)
    a = F.pad(a, (0, (n - 1) * stride + kernel_size - length))
    return torch.stack([a[..., i * stride:i * stride + kernel_size] for i in range(n)], dim=-2)


def center_trim(t, length):
    """src/model/demucs.py:34-50"""
    delta = t.shape[-1] - length
    if delta < 0:
        raise ValueError(f"tensor must be larger than reference. Delta is {delta}.")
    return t[..., delta // 2:t.shape[-1] - (delta - delta // 2)] if delta else t


def lstm_direction(x, w_ih, w_hh, b_ih, b_hh, reverse):
    """One direction of one nn.LSTM layer, zero initial state; x [T, B, in] -> [T, B, H]; gate order i, f, g, o."""
    T, B, _ = x.shape
    H = w_hh.shape[1]
    pre = x @ w_ih.t() + (b_ih + b_hh)
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    out = [None] * T
    for t in (range(T - 1, -1, -1) if reverse else range(T)):
        i, f, g, o = (pre[t] + h @ w_hh.t()).chunk(4, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h = torch.sigmoid(o) * torch.tanh(c)
        out[t] = h
    return torch.stack(out)


def blstm(x, p, pre, layers, max_steps, skip):
    """BLSTM (src/model/demucs.py:73-120): x [B, C, T]; bidirectional nn.LSTM(C, C, layers) + Linear(2C, C), applied on
    overlapping chunks of `max_steps` (hop max_steps/2) when T > max_steps, the middle half of each chunk kept."""
    B, C, T = x.shape
    y = x
    framed = max_steps is not None and T > max_steps
    if framed:
        width, stride = max_steps, max_steps // 2
        frames = unfold(x, width, stride)                                      # [B, C, nframes, width]
        nframes = frames.shape[2]
        x = frames.permute(0, 2, 1, 3).reshape(-1, C, width)
    h = x.permute(2, 0, 1)
    for l in range(layers):
        q = f"{pre}lstm."
        fw = lstm_direction(h, p[f"{q}weight_ih_l{l}"], p[f"{q}weight_hh_l{l}"], p[f"{q}bias_ih_l{l}"], p[f"{q}bias_hh_l{l}"], False)
        bw = lstm_direction(h, p[f"{q}weight_ih_l{l}_reverse"], p[f"{q}weight_hh_l{l}_reverse"], p[f"{q}bias_ih_l{l}_reverse"],
                            p[f"{q}bias_hh_l{l}_reverse"], True)
        h = torch.cat([fw, bw], dim=2)
    h = F.linear(h, p[pre + "linear.weight"], p[pre + "linear.bias"]).permute(1, 2, 0)
    if framed:
        fr = h.reshape(B, -1, C, width)
        limit = stride // 2
        out = []
        for k in range(nframes):
            if k == 0:
                out.append(fr[:, k, :, :-limit])
            elif k == nframes - 1:
                out.append(fr[:, k, :, limit:])
            else:
                out.append(fr[:, k, :, limit:-limit])
        h = torch.cat(out, -1)[..., :T]
    return h + y if skip else h


def local_state(x, p, pre, heads, ndecay):
    """LocalState with nfreqs = 0 (src/model/demucs.py:210-269): softmax over the KEY axis of content-based scores plus a
    learnt distance penalty, the diagonal set to -100."""
    B, C, T = x.shape
    conv = lambda n: F.conv1d(x, p[f"{pre}{n}.weight"], p[f"{pre}{n}.bias"])
    idx = torch.arange(T, dtype=x.dtype)
    delta = idx[:, None] - idx[None, :]                                        # [t (keys), s (queries)]
    q = conv("query").view(B, heads, -1, T)
    k = conv("key").view(B, heads, -1, T)
    dots = torch.einsum("bhct,bhcs->bhts", k, q) / k.shape[2] ** 0.5
    if ndecay:
        decays = torch.arange(1, ndecay + 1, dtype=x.dtype)
        dq = torch.sigmoid(conv("query_decay").view(B, heads, -1, T)) / 2
        kern = -decays.view(-1, 1, 1) * delta.abs() / ndecay ** 0.5
        dots = dots + torch.einsum("fts,bhfs->bhts", kern, dq)
    dots = dots.masked_fill(torch.eye(T, dtype=torch.bool), -100.0)
    w = torch.softmax(dots, dim=2)
    content = conv("content").view(B, heads, -1, T)
    res = torch.einsum("bhts,bhct->bhcs", w, content).reshape(B, -1, T)
    return x + F.conv1d(res, p[pre + "proj.weight"], p[pre + "proj.bias"])


def dconv_layout(attn, lstm):
    """Sequential indices inside one DConv layer (src/model/demucs.py:190-201): conv 0, norm 1, act 2, [BLSTM 3], [LocalState
    3 or 4], 1x1 conv, norm, GLU, LayerScale."""
    i = 3
    out = {}
    if lstm:
        out["lstm"] = i; i += 1
    if attn:
        out["attn"] = i; i += 1
    out["conv2"], out["norm2"], out["scale"] = i, i + 1, i + 3
    return out


def dconv(x, p, pre, cfg: DemucsConfig, attn, lstm, taps=None):
    """DConv (src/model/demucs.py:139-207) with norm=True, gelu=True, kernel 3, heads 4, ndecay 4 (the defaults the
    Demucs constructor leaves in place, :393-394)."""
    lay = dconv_layout(attn, lstm)
    act = F.gelu if cfg.gelu else F.relu
    for d in range(abs(cfg.dconv_depth)):
        q = f"{pre}layers.{d}."
        dil = 2 ** d if cfg.dconv_depth > 0 else 1
        h = F.conv1d(x, p[q + "0.weight"], p[q + "0.bias"], dilation=dil, padding=dil)
        h = act(F.group_norm(h, 1, p[q + "1.weight"], p[q + "1.bias"]))
        if lstm:
            h = blstm(h, p, f"{q}{lay['lstm']}.", 2, 200, True)
        if attn:
            h = local_state(h, p, f"{q}{lay['attn']}.", 4, 4)
        h = F.conv1d(h, p[f"{q}{lay['conv2']}.weight"], p[f"{q}{lay['conv2']}.bias"])
        h = F.glu(F.group_norm(h, 1, p[f"{q}{lay['norm2']}.weight"], p[f"{q}{lay['norm2']}.bias"]), dim=1)
        x = x + p[f"{q}{lay['scale']}.scale"][:, None] * h
    return x


def demucs_forward(p, mix, cfg: DemucsConfig, taps=None):
    """mix [B, ac, T] -> [B, S, ac, T] (src/model/demucs.py:453-490)."""
    x = mix
    length = x.shape[-1]
    if cfg.normalize:
        mono = mix.mean(dim=1, keepdim=True)
        mean = mono.mean(dim=-1, keepdim=True)
        std = mono.std(dim=-1, keepdim=True)
        x = (x - mean) / (1e-5 + std)
    else:
        mean, std = 0, 1
    delta = cfg.valid_length(length) - length
    x = F.pad(x, (delta // 2, delta - delta // 2))
    if cfg.resample:
        x = resample_frac(x, 1, 2)
    act2 = F.gelu if cfg.gelu else F.relu
    act = (lambda t: F.glu(t, dim=1)) if cfg.glu else F.relu
    saved = []
    for i in range(cfg.depth):
        q = f"encoder.{i}."
        norm = (lambda t, n: F.group_norm(t, cfg.norm_groups, p[n + "weight"], p[n + "bias"])) if i >= cfg.norm_starts else (lambda t, n: t)
        x = F.conv1d(x, p[q + "0.weight"], p[q + "0.bias"], stride=cfg.stride)
        x = act2(norm(x, q + "1."))
        j = 3
        if cfg.dconv_mode & 1:
            x = dconv(x, p, f"{q}{j}.", cfg, i >= cfg.dconv_attn, i >= cfg.dconv_lstm)
            j += 1
        if cfg.rewrite:
            x = F.conv1d(x, p[f"{q}{j}.weight"], p[f"{q}{j}.bias"])
            x = act(norm(x, f"{q}{j + 1}."))
        saved.append(x)
        if taps is not None:
            taps[f"enc{i}"] = x
    if cfg.lstm_layers:
        x = blstm(x, p, "lstm.", cfg.lstm_layers, None, False)
    for jdx in range(cfg.depth):
        index = cfg.depth - 1 - jdx
        q = f"decoder.{jdx}."
        norm = (lambda t, n: F.group_norm(t, cfg.norm_groups, p[n + "weight"], p[n + "bias"])) if index >= cfg.norm_starts else (lambda t, n: t)
        skip = center_trim(saved.pop(-1), x.shape[-1])
        x = x + skip
        j = 0
        if cfg.rewrite:
            x = F.conv1d(x, p[q + "0.weight"], p[q + "0.bias"], padding=cfg.context)
            x = act(norm(x, q + "1."))
            j = 3
        if cfg.dconv_mode & 2:
            x = dconv(x, p, f"{q}{j}.", cfg, index >= cfg.dconv_attn, index >= cfg.dconv_lstm)
            j += 1
        x = F.conv_transpose1d(x, p[f"{q}{j}.weight"], p[f"{q}{j}.bias"], stride=cfg.stride)
        if index > 0:
            x = act2(norm(x, f"{q}{j + 1}."))
        if taps is not None:
            taps[f"dec{jdx}"] = x
    if cfg.resample:
        x = resample_frac(x, 2, 1)
    x = x * std + mean
    x = center_trim(x, length)
    return x.view(x.size(0), cfg.S, cfg.audio_channels, x.size(-1))


def param_shapes(cfg: DemucsConfig):
    """[(state_dict name, shape)] in the reference's parameters() order."""
    out = []
    K, ctx = cfg.kernel_size, cfg.context
    cs = 2 if cfg.glu else 1

    def conv(n, co, ci, k):
        out.extend([(n + ".weight", (co, ci, k)), (n + ".bias", (co,))])

    def gn(n, c):
        out.extend([(n + ".weight", (c,)), (n + ".bias", (c,))])

    def dconv_params(pre, ch, attn, lstm):
        hid = int(ch / cfg.dconv_comp)
        lay = dconv_layout(attn, lstm)
        for d in range(abs(cfg.dconv_depth)):
            q = f"{pre}.layers.{d}."
            conv(q + "0", hid, ch, 3); gn(q + "1", hid)
            if lstm:
                b = f"{q}{lay['lstm']}."
                for l in range(2):
                    for sfx in ("", "_reverse"):
                        out.extend([(f"{b}lstm.weight_ih_l{l}{sfx}", (4 * hid, hid if l == 0 else 2 * hid)),
                                    (f"{b}lstm.weight_hh_l{l}{sfx}", (4 * hid, hid)),
                                    (f"{b}lstm.bias_ih_l{l}{sfx}", (4 * hid,)), (f"{b}lstm.bias_hh_l{l}{sfx}", (4 * hid,))])
                out.extend([(b + "linear.weight", (hid, 2 * hid)), (b + "linear.bias", (hid,))])
            if attn:
                a = f"{q}{lay['attn']}."
                conv(a + "content", hid, hid, 1); conv(a + "query", hid, hid, 1); conv(a + "key", hid, hid, 1)
                conv(a + "query_decay", 16, hid, 1); conv(a + "proj", hid, hid, 1)
            conv(f"{q}{lay['conv2']}", 2 * ch, hid, 1); gn(f"{q}{lay['norm2']}", 2 * ch)
            out.append((f"{q}{lay['scale']}.scale", (ch,)))

    enc, dec = [], []
    for i, (cin, ch) in enumerate(cfg.layer_channels()):
        mark = len(out)
        q = f"encoder.{i}"
        normed = i >= cfg.norm_starts
        conv(q + ".0", ch, cin, K)
        if normed:
            gn(q + ".1", ch)
        j = 3
        if cfg.dconv_mode & 1:
            dconv_params(f"{q}.{j}", ch, i >= cfg.dconv_attn, i >= cfg.dconv_lstm); j += 1
        if cfg.rewrite:
            conv(f"{q}.{j}", cs * ch, ch, 1)
            if normed:
                gn(f"{q}.{j + 1}", cs * ch)
        enc.append(out[mark:]); del out[mark:]
        q = f"decoder.{cfg.depth - 1 - i}"
        cout = cin if i > 0 else cfg.S * cfg.audio_channels
        j = 0
        if cfg.rewrite:
            conv(q + ".0", cs * ch, ch, 2 * ctx + 1)
            if normed:
                gn(q + ".1", cs * ch)
            j = 3
        if cfg.dconv_mode & 2:
            dconv_params(f"{q}.{j}", ch, i >= cfg.dconv_attn, i >= cfg.dconv_lstm); j += 1
        out.extend([(f"{q}.{j}.weight", (ch, cout, K)), (f"{q}.{j}.bias", (cout,))])
        if i > 0 and normed:
            gn(f"{q}.{j + 1}", cout)
        dec.append(out[mark:]); del out[mark:]
    for e in enc:
        out.extend(e)
    for d in reversed(dec):
        out.extend(d)
    return out


def is_trainable(key: str) -> bool:
    return True   # Demucs has parameters only (GroupNorm: no running statistics)
