"""Drop-in counterparts of the reference's ``models`` package (brainformer, gpt2_model) + the notebook-only classes."""
from . import brainformer, gpt2_model, notebook_models, simple_mae, vq_brain  # noqa: F401
