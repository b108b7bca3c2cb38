from .base import *  # noqa: F401,F403
from .cnn import *  # noqa: F401,F403
from .rnn import *  # noqa: F401,F403
