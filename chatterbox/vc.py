from chatterbox_amd.api import ChatterboxVC  # noqa: F401
