from .conv import *      # noqa: F401,F403
from .linear import *    # noqa: F401,F403
from .matmul import *    # noqa: F401,F403
