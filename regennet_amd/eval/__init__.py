from .auto_regressive import sample_auto_regressive  # noqa: F401
from .fid import calculate_activation_statistics, calculate_fid, calculate_frechet_distance  # noqa: F401
from .metrics import Evaluation, calculate_accuracy, calculate_diversity_multimodality  # noqa: F401
from .stgcn import STGCN  # noqa: F401
