"""The data path in front of Solver.train, on the device.

The reference normalises and crops every utterance in WavDataset.__getitem__ on CPU workers (src/dataset.py:95-170: z-score
:147-152, linear-scale :154-160, sample_fixed_length_data_aligned src/utils.py:63-87) and then cuts / pads / stacks the batch in
collate_fn_pad (src/distrib.py:38-98).  DeviceBatcher takes the RAW utterances of a batch (what sf.read returned, file I/O stays on
the host), moves them to the GPU as one flat buffer and builds the reference's batch tuple with two kernels
(csrc/data.hip: row statistics; normalise + crop + segment + batch-concatenate in one pass).

    batcher = DeviceBatcher(config.dset, normalize="z-score", sample_length=64000, drop_last=True)
    mixture, sources, mix_meta, src_meta, names, index_batch = batcher(items)      # items: [(mixture [C, n], sources [S, C, n], name)]

The tuple is the one `for batch in dataloader` yields in src/solver.py:430-441; `Solver._run_one_epoch` consumes it unchanged.
The crop offsets come from numpy's global RNG in the reference's order (one np.random.randint per utterance), so a seeded run
crops where the reference crops."""
import numpy as np
import torch

from ._lib import SehipError, call, ptr, stream, require_gpu

_MODES = {"": 0, None: 0, "none": 0, "z-score": 1, "linear-scale": 2}
EPS = 1e-6   # src/dataset.py:146


def plan_batch(lengths, seg, sample_length, drop_last, starts):
    """Host arithmetic of the crop (src/utils.py:63-87) and of collate_fn_pad's cut / pad (src/distrib.py:55-66) for utterances of
    `lengths` samples.  Returns per utterance (start, valid samples, number of segments)."""
    out = []
    for n, st in zip(lengths, starts):
        if sample_length:
            L = sample_length
            valid = min(n - st, sample_length) if n > sample_length else n      # a short utterance is zero-padded to the sample
        else:
            L, valid = n, n
        if L < seg:
            L = seg
        if L % seg:
            L = seg * (L // seg) if drop_last else (L // seg + 1) * seg
        out.append((st, min(valid, L), L // seg))
    return out


class DeviceBatcher:
    def __init__(self, config, normalize="", sample_length=0, drop_last=True, device="cuda", rng=None):
        if normalize not in _MODES:
            raise SehipError(f"DeviceBatcher: normalize must be one of {[k for k in _MODES if k]}, got {normalize!r}")
        self.seg = int(config.segment * config.sample_rate)
        self.mode, self.sample_length, self.drop_last = _MODES[normalize], int(sample_length or 0), bool(drop_last)
        self.device = torch.device(device)
        self.rng = rng if rng is not None else np.random
        if self.seg <= 0:
            raise SehipError("DeviceBatcher: config.segment * config.sample_rate must be positive")

    def draw_starts(self, lengths):
        """one np.random.randint per utterance, in order (src/utils.py:76-77; a short utterance draws randint(1) == 0)"""
        if not self.sample_length:
            return [0] * len(lengths)
        return [int(self.rng.randint(max(n, self.sample_length) - self.sample_length + 1)) for n in lengths]

    def __call__(self, items, starts=None):
        if not items:
            raise SehipError("DeviceBatcher: empty batch")
        mixes = [torch.as_tensor(it[0], dtype=torch.float32) for it in items]
        srcs = [torch.as_tensor(it[1], dtype=torch.float32) for it in items]
        names = [it[2] if len(it) > 2 else None for it in items]
        C, S = mixes[0].shape[0], srcs[0].shape[0]
        for m, s_ in zip(mixes, srcs):
            if m.dim() != 2 or s_.dim() != 3 or m.shape[0] != C or tuple(s_.shape[:2]) != (S, C) or s_.shape[-1] != m.shape[-1]:
                raise SehipError(f"DeviceBatcher: mixture [C, n] / sources [S, C, n] expected, got {tuple(m.shape)} / {tuple(s_.shape)}")
        lengths = [int(m.shape[-1]) for m in mixes]
        if starts is None:
            starts = self.draw_starts(lengths)
        plan = plan_batch(lengths, self.seg, self.sample_length, self.drop_last, starts)
        G = sum(p[2] for p in plan)
        if G == 0:
            raise SehipError("DeviceBatcher: no segment survives drop_last")
        rpi = C + S * C                                             # raw rows per utterance: mixture channels, then (source, channel)
        R = len(items) * rpi
        row_off = np.zeros(R + 1, dtype=np.int64)
        row_off[1:] = np.cumsum(np.repeat(lengths, rpi))
        flat = torch.empty(int(row_off[-1]), dtype=torch.float32).pin_memory() if torch.cuda.is_available() else torch.empty(int(row_off[-1]))
        for i, (m, s_) in enumerate(zip(mixes, srcs)):
            n, b0 = lengths[i], int(row_off[i * rpi])
            flat[b0:b0 + C * n].view(C, n).copy_(m)
            flat[b0 + C * n:b0 + rpi * n].view(S, C, n).copy_(s_)
        # output rows: mixture [G, C, seg] first, then sources [G, S, C, seg]
        n_mix, n_src = G * C, G * S * C
        out_row = np.empty(n_mix + n_src, dtype=np.int32)
        out_start = np.empty(n_mix + n_src, dtype=np.int64)
        out_valid = np.empty(n_mix + n_src, dtype=np.int32)
        g = 0
        for i, (st, valid, nseg) in enumerate(plan):
            for k in range(nseg):
                v = max(0, min(self.seg, valid - k * self.seg))
                a = (g + k) * C
                out_row[a:a + C] = i * rpi + np.arange(C)
                out_start[a:a + C], out_valid[a:a + C] = st + k * self.seg, v
                b_ = n_mix + (g + k) * S * C
                out_row[b_:b_ + S * C] = i * rpi + C + np.arange(S * C)
                out_start[b_:b_ + S * C], out_valid[b_:b_ + S * C] = st + k * self.seg, v
            g += nseg
        dev = self.device
        require_gpu(torch.empty(0, device=dev), "DeviceBatcher")
        raw = flat.to(dev, non_blocking=True)
        d_off = torch.from_numpy(row_off).to(dev)
        stats = torch.zeros(R, 4, dtype=torch.float32, device=dev)
        if self.mode:
            call("sehip_wav_row_stats", ptr(raw), ptr(d_off), R, ptr(stats), stream())
        out = torch.empty((n_mix + n_src) * self.seg, dtype=torch.float32, device=dev)
        # (named, so that the three index tables stay allocated until the launch has been enqueued behind their copies)
        d_row, d_start, d_valid = torch.from_numpy(out_row).to(dev), torch.from_numpy(out_start).to(dev), torch.from_numpy(out_valid).to(dev)
        call("sehip_wav_collate", ptr(raw), ptr(d_off), ptr(d_row), ptr(d_start), ptr(d_valid), ptr(stats), self.mode, EPS, self.seg,
             n_mix + n_src, ptr(out), stream())
        mixture = out[:n_mix * self.seg].view(G, C, self.seg)
        sources = out[n_mix * self.seg:].view(G, S, C, self.seg)
        zero = 0
        mix_meta, src_meta = [], []
        for i in range(len(items)):          # the reference's per-utterance dictionaries (src/dataset.py:131-143), values on the device
            sm = stats[i * rpi:i * rpi + C]
            ss = stats[i * rpi + C:(i + 1) * rpi].view(S, C, 4)
            if self.mode == 1:
                mix_meta.append({"min": zero, "max": zero, "mean": sm[:, 0:1], "std": sm[:, 1:2]})
                src_meta.append({"min": zero, "max": zero, "mean": ss[..., 0:1], "std": ss[..., 1:2]})
            elif self.mode == 2:
                mix_meta.append({"min": sm[:, 2:3], "max": sm[:, 3:4], "mean": zero, "std": zero})
                src_meta.append({"min": ss[..., 2:3], "max": ss[..., 3:4], "mean": zero, "std": zero})
            else:
                mix_meta.append({"min": zero, "max": zero, "mean": zero, "std": zero})
                src_meta.append({"min": zero, "max": zero, "mean": zero, "std": zero})
        return mixture, sources, mix_meta, src_meta, names, [p[2] for p in plan]
