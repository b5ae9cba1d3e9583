"""Device-side input pipeline with the reference's `utils/data_utils.py` names.

Reference map: process_signal utils/data_utils.py:115-155 (channel concat, block-wise z-score with std == 0 -> 1, Gaussian
smoothing sigma = 1 over time), z_score_per_block_scaling :78-109 (sklearn StandardScaler per block = the same statistics),
pad_truncate_brain_list :243-267, MAX_INPUT_LEN / MAX_TOKENS :11-12, pad_token_list :282-286.  The statistics, normalisation,
smoothing and padding run as two HIP kernels (fk_block_stats, fk_zscore_smooth_pad) over trials packed row-wise in HBM; the
functions keep the reference's list-of-arrays signatures and add `process_and_pad`, which returns the padded batch tensor the
training loop consumes without a host round trip.  .mat file parsing and the tokenizer stay host-side glue (out of scope)."""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch

from .. import kernels as K
from .._lib import call, lib

MAX_INPUT_LEN = 768
MAX_TOKENS = 25

# recording sessions of the T12 data set in chronological order (the reference's DATE_TO_INDEX, utils/data_utils.py:14-38)
_DATES = ["04.28", "05.05", "05.17", "05.19", "05.24", "05.26", "06.02", "06.07", "06.14", "06.16", "06.21", "06.23", "06.28",
          "07.05", "07.14", "07.21", "07.27", "07.29", "08.02", "08.11", "08.13", "08.18", "08.23", "08.25"]
DATE_TO_INDEX = {f"t12.2022.{d}": i for i, d in enumerate(_DATES)}


def _pack(brain_list: Sequence[np.ndarray], block_list, device):
    lens = [int(b.shape[0]) for b in brain_list]
    C = int(brain_list[0].shape[1])
    assert all(b.ndim == 2 and b.shape[1] == C for b in brain_list), "every trial must be [time, channels]"
    off = np.zeros(len(lens) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    x = torch.from_numpy(np.concatenate([np.asarray(b, dtype=np.float32) for b in brain_list], axis=0)).to(device)
    uniq, inv = np.unique(np.asarray(block_list), return_inverse=True)
    return x, torch.from_numpy(off).to(device), torch.from_numpy(inv.astype(np.int32)).to(device), lens, C, len(uniq)


def block_stats(x: torch.Tensor, off: torch.Tensor, block: torch.Tensor, nblocks: int):
    """(mean, std) [nblocks, C] fp32 of the packed trials x [rows, C]; std == 0 -> 1."""
    n, C = off.numel() - 1, x.shape[1]
    mean = torch.empty((nblocks, C), dtype=torch.float32, device=x.device)
    std = torch.empty_like(mean)
    nb = lib().fk_block_stats_workspace_bytes(n, C)
    ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
    call("fk_block_stats", x.data_ptr(), off.data_ptr(), block.data_ptr(), n, C, nblocks, mean.data_ptr(), std.data_ptr(),
         ws.data_ptr(), nb, K._stream())
    return mean, std


def process_and_pad(brain_list: Sequence[np.ndarray], block_list, max_length: int = MAX_INPUT_LEN, sigma: float = 1.0,
                    device="cuda") -> torch.Tensor:
    """[n_trials, max_length, C] fp32 on `device`: block-wise z-score + Gaussian smoothing + zero-pad / truncate."""
    x, off, blk, lens, C, nblocks = _pack(brain_list, block_list, device)
    mean, std = block_stats(x, off, blk, nblocks)
    out = torch.empty((len(lens), max_length, C), dtype=torch.float32, device=x.device)
    call("fk_zscore_smooth_pad", x.data_ptr(), off.data_ptr(), blk.data_ptr(), mean.data_ptr(), std.data_ptr(), out.data_ptr(),
         len(lens), C, max_length, float(sigma), K._stream())
    return out


def process_signal(voltage_list, spikes_list, block_list) -> np.ndarray:
    """Reference signature (:115): object array of per-trial [time, 2 x channels] arrays, z-scored per block and smoothed."""
    brain = [np.concatenate([v, s], axis=1) for v, s in zip(voltage_list, spikes_list)]
    tmax = max(b.shape[0] for b in brain)
    dense = process_and_pad(brain, np.asarray(block_list), max_length=tmax).cpu().numpy()
    out = np.empty(len(brain), dtype=object)
    for i, b in enumerate(brain):
        out[i] = dense[i, :b.shape[0]]
    return out


def z_score_per_block_scaling(brain_list, idx_list) -> List[np.ndarray]:
    """Reference signature (:78): StandardScaler per block, no smoothing."""
    x, off, blk, lens, C, nblocks = _pack(brain_list, idx_list, "cuda")
    mean, std = block_stats(x, off, blk, nblocks)
    z = ((x - mean[blk.long()].repeat_interleave(torch.tensor(lens, device=x.device), dim=0))
         / std[blk.long()].repeat_interleave(torch.tensor(lens, device=x.device), dim=0)).cpu().numpy()
    o = off.cpu().numpy()
    return [z[o[i]:o[i + 1]] for i in range(len(lens))]


def pad_truncate_brain_list(brain_list, max_length):
    """Reference signature (:243): host lists in, host lists out (use process_and_pad for the fused device path)."""
    out = []
    for b in brain_list:
        t = b.shape[0]
        out.append(b[:max_length] if t > max_length else np.pad(b, ((0, max_length - t), (0, 0)), mode="constant"))
    return out


def pad_token_list(token_list, max_tokens):
    n = max_tokens - len(token_list)
    if n > 0:
        token_list.extend([-100] * n)
    return token_list


def remove_padding(token_list):
    return [t for t in token_list if t != -100]


# ------------------------------------------------------------------------------------------------ host-side glue
# File parsing, text clean-up and the dataset wrapper are host code in the reference too (utils/data_utils.py:40-76,159-228,230-241,
# 270-281,291-344); only their signal processing runs on the device (above).
def min_max_per_block_scaling(brain_list, idx_list):
    """Per-block MinMaxScaler (:44-75): (x - min) / (max - min) with the block's per-channel extrema (constant channels -> 0)."""
    blocks = {}
    for i, b in enumerate(idx_list):
        blocks.setdefault(b, []).append(i)
    out = [None] * len(brain_list)
    for members in blocks.values():
        cat = np.concatenate([brain_list[i] for i in members])
        lo, span = cat.min(axis=0), cat.max(axis=0) - cat.min(axis=0)
        span = np.where(span == 0, 1.0, span)
        for i in members:
            out[i] = (brain_list[i] - lo) / span
    return out


def process_text(arr):
    return [t.strip() for t in arr]


def process_file(data_file):
    """One session .mat file -> (z-scored spike-power trials, sentences, dates) like :162-187 (z-score on the device)."""
    import scipy.io
    data = scipy.io.loadmat(data_file)
    n_trials = data["blockIdx"].shape[0]
    voltage_list = data["spikePow"][0][:]
    block_list = data["blockIdx"][:, 0]
    brain_list = z_score_per_block_scaling(list(voltage_list), list(block_list))
    return brain_list, process_text(data["sentenceText"]), [data_file.stem] * n_trials


def process_all_files(path):
    data = {"brain_list": [], "sentence_list": [], "date_list": []}
    for f in sorted(path.glob("*.mat")):
        brains, sentences, dates = process_file(f)
        data["brain_list"].extend(brains)
        data["sentence_list"].extend(sentences)
        data["date_list"].extend(dates)
    return data


def remove_punctuation(text):
    import string
    drop = set(string.punctuation) - {"'"}
    return "".join(ch for ch in text if ch not in drop)


def process_string(text):
    return remove_punctuation(text.lower())


def save_sentences_to_txt(fpath, sentences, string_processing_fn):
    with open(fpath, "w", encoding="utf-8") as f:
        f.writelines(string_processing_fn(s_) + "\n" for s_ in sentences)


def load_sentences_from_txt(fpath):
    with open(fpath, "r", encoding="utf-8") as f:
        return [line.strip() for line in f]


def find_long_samples(sample_list, max_length):
    return [i for i, s_ in enumerate(sample_list) if len(s_) > max_length]


def get_tokenizer(tokenizer):
    bos, eos = tokenizer.bos_token, tokenizer.eos_token

    def tokenize_txt(text):
        return tokenizer(bos + text + eos).input_ids
    return tokenize_txt


class BrainDataset(torch.utils.data.Dataset):
    """(input [768, C] float32, padded token targets, date) per trial, as the reference's dataset (:291-344)."""

    def __init__(self, path, tokenize_function=None):
        print("Runed processing of the ", path)
        data = process_all_files(path)
        self.inputs, self.targets, self.date = data["brain_list"], data["sentence_list"], data["date_list"]
        self.date_to_index = DATE_TO_INDEX
        if tokenize_function is not None:
            self.targets_tokens = [np.asarray(pad_token_list(tokenize_function(t), MAX_TOKENS), dtype=np.int64) for t in self.targets]
        else:
            self.targets_tokens = self.targets[:]
        self.inputs = pad_truncate_brain_list(self.inputs, MAX_INPUT_LEN)

    def __len__(self) -> int:
        return len(self.inputs)

    def remove_bad_samples(self, bad_indices):
        for i in reversed(bad_indices):
            del self.inputs[i], self.targets[i], self.targets_tokens[i], self.date[i]

    def __getitem__(self, idx: int):
        return self.inputs[idx].astype(np.float32), self.targets_tokens[idx], self.date[idx]
