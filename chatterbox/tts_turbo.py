from chatterbox_amd.api import ChatterboxTurboTTS, Conditionals  # noqa: F401
from chatterbox_amd.text import punc_norm_turbo as punc_norm  # noqa: F401
