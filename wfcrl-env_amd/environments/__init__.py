from .data_cases import FarmCase, FastFarmCase, FlorisCase  # noqa: F401
from .registration import list_envs, make  # noqa: F401
