"""Import-path compatibility with the reference package: `from chatterbox.tts import ChatterboxTTS`,
`from chatterbox.mtl_tts import ChatterboxMultilingualTTS, SUPPORTED_LANGUAGES`, `from chatterbox.vc import ChatterboxVC`
resolve to the MI355X implementation in chatterbox_amd (reference src/chatterbox/__init__.py:9-10)."""
from chatterbox_amd.api import (ChatterboxMultilingualTTS, ChatterboxTTS, ChatterboxTurboTTS, ChatterboxVC,  # noqa: F401
                                SUPPORTED_LANGUAGES)
