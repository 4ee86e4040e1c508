from .base import *            # noqa: F401,F403  (Operator, register_operator, get_operator)
from .blackhole import *       # noqa: F401,F403
from .dynamic_mri import *     # noqa: F401,F403
from .latent_wrapper import LatentWrapper   # noqa: F401
from .radial_mri import *      # noqa: F401,F403
