from chatterbox_amd.api import ChatterboxMultilingualTTS, Conditionals, SUPPORTED_LANGUAGES, T3Cond  # noqa: F401
from chatterbox_amd.text import punc_norm  # noqa: F401
