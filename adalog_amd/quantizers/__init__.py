from .uniform import *      # noqa: F401,F403
from .logarithm import *    # noqa: F401,F403
from .uniform import UniformQuantizer
from .logarithm import AdaLogQuantizer, ShiftAdaLogQuantizer
