"""recmv.utils — hot-path helpers of the reference's `utils` package (utils/__init__.py:1-2)."""
from .utils import *          # noqa: F401,F403
from .FindSurfacePs import *  # noqa: F401,F403
