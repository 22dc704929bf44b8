from .dccrn import DCCRN  # noqa: F401
from . import types  # noqa: F401
