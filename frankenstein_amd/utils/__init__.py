"""Drop-in counterpart of the reference's ``utils`` package (train_utils)."""
from . import train_utils  # noqa: F401
