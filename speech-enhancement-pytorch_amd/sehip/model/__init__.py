from .dccrn import DCCRN  # noqa: F401
from . import types  # noqa: F401
from .dnn import DeepNeuralNetwork  # noqa: F401
from .dcunet import DCUnet  # noqa: F401
from .conv_tasnet import ConvTasNet  # noqa: F401
from .demucs import Demucs  # noqa: F401
