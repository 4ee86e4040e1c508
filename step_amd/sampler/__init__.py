from .base import *      # noqa: F401,F403
from .daps import *      # noqa: F401,F403
