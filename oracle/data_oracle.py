"""TEST INFRASTRUCTURE (CPU oracle; never imported by the product path).

Restatement of the reference's data path between sf.read and the Solver:
  normalise          src/dataset.py:145-160   z-score with torch.std (unbiased), eps 1e-6; linear-scale as evidently intended
                                               (the reference subtracts torch.max's (values, indices) tuple and raises)
  crop               src/utils.py:63-87       sample_fixed_length_data_aligned: zero pad_last to the sample length, window [start, end)
  collate            src/distrib.py:38-98     collate_fn_pad
Pinned by tests/golden/collate_cases.npz: oracle/gen_golden_collate.py runs the imported collate_fn_pad and
sample_fixed_length_data_aligned on utterances normalised by the three z-score lines of WavDataset.__getitem__ (that method itself
needs soundfile, which is absent here)."""
import torch
import torch.nn.functional as F


def pad_last(t, size):
    return F.pad(t, (0, size))


def normalise(mixture, sources, mode, eps=1e-6):
    if mode == "z-score":
        mixture = (mixture - mixture.mean(-1, keepdim=True)) / (mixture.std(-1, keepdim=True) + eps)
        sources = (sources - sources.mean(-1, keepdim=True)) / (sources.std(-1, keepdim=True) + eps)
    elif mode == "linear-scale":
        lo, hi = mixture.min(-1, keepdim=True).values, mixture.max(-1, keepdim=True).values
        mixture = (mixture - lo) / (hi - lo + eps)
        lo, hi = sources.min(-1, keepdim=True).values, sources.max(-1, keepdim=True).values
        sources = (sources - lo) / (hi - lo + eps)
    return mixture, sources


def crop(data_list, sample_length, start):
    if data_list[0].shape[-1] <= sample_length:
        data_list = [pad_last(d, sample_length - d.shape[-1]) for d in data_list]
    return [d[..., start:start + sample_length] for d in data_list]


def collate(items, segment_length, drop_last=True):
    """items: [(mixture [C, n], sources [S, C, n])] -> (mixture [G, C, seg], sources [G, S, C, seg], segments per utterance)"""
    bm, bs, idx = [], [], []
    for mixture, sources in items:
        if mixture.shape[-1] < segment_length:
            mixture = pad_last(mixture, segment_length - mixture.shape[-1])
            sources = pad_last(sources, segment_length - sources.shape[-1])
        if mixture.shape[-1] % segment_length:
            if drop_last:
                keep = segment_length * (mixture.shape[-1] // segment_length)
                mixture, sources = mixture[..., :keep], sources[..., :keep]
            else:
                to = (mixture.shape[-1] // segment_length + 1) * segment_length
                mixture, sources = pad_last(mixture, to - mixture.shape[-1]), pad_last(sources, to - sources.shape[-1])
        c, length = mixture.shape
        nseg = length // segment_length
        bm.append(mixture.reshape(c, nseg, segment_length))
        bs.append(sources.reshape(sources.shape[0], c, nseg, segment_length))
        idx.append(nseg)
    return torch.cat(bm, 1).permute(1, 0, 2), torch.cat(bs, 2).permute(2, 0, 1, 3), idx
