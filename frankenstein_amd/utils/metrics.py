"""Host-side sequence metrics for the inference path.

The reference's only WER is HuggingFace ``evaluate.load("wer")`` in the Whisper notebook (notebooks/whisper_hugging_face.ipynb, the
``compute_metrics`` cell): 100 x (substitutions + deletions + insertions over all samples) / (reference words over all samples), words
split on whitespace.  ``wer`` below is that definition (as a fraction); ``token_error_rate`` is the same on token-id sequences, which is
what the parity tests use to compare decoded sequences of this build with the CPU reference ("WER 0" = identical decodes)."""
from __future__ import annotations

from typing import Iterable, Sequence


def edit_distance(ref: Sequence, hyp: Sequence) -> int:
    """Levenshtein distance (unit costs) between two sequences, O(len(ref) x len(hyp)) time, O(len(hyp)) memory."""
    prev = list(range(len(hyp) + 1))
    for i, r in enumerate(ref, 1):
        cur = [i] + [0] * len(hyp)
        for j, h in enumerate(hyp, 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (r != h))
        prev = cur
    return prev[-1]


def _rate(refs: Iterable[Sequence], hyps: Iterable[Sequence]) -> float:
    errors = total = 0
    refs, hyps = list(refs), list(hyps)
    if len(refs) != len(hyps):
        raise ValueError(f"{len(refs)} references vs {len(hyps)} hypotheses")
    for r, h in zip(refs, hyps):
        errors += edit_distance(r, h)
        total += len(r)
    if total == 0:
        raise ValueError("the references are empty")
    return errors / total


def wer(references: Iterable[str], predictions: Iterable[str]) -> float:
    """Word error rate over a corpus: total word-level edit distance / total reference words (a fraction; x100 for per cent)."""
    return _rate([r.split() for r in references], [p.split() for p in predictions])


def token_error_rate(references: Iterable[Sequence[int]], predictions: Iterable[Sequence[int]], ignore_index: int = -100) -> float:
    """The same rate on token-id sequences; ``ignore_index`` entries (the reference's label padding) are dropped first."""
    strip = lambda s: [int(t) for t in s if int(t) != ignore_index]
    return _rate([strip(r) for r in references], [strip(p) for p in predictions])
