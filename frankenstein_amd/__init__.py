"""frankenstein_amd — MI355X-native (gfx950) hot path of ALVI-Labs/frankenstein: the brainformer / GPT-2
decoder training step as hand-written HIP kernels behind a C ABI (include/franken_hip.h), with the
reference's Python module surface on top (frankenstein_amd.models.brainformer / .gpt2_model,
frankenstein_amd.utils.train_utils)."""
from .engine import bump_weight_epoch, compute_dtype, set_compute_dtype  # noqa: F401

__all__ = ["set_compute_dtype", "compute_dtype", "bump_weight_epoch"]
