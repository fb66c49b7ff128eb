"""takzero_amd — MI355X-native engine for takzero's self-play / reanalyze hot path.

The package is a thin host-side mirror of the reference's Env/Agent/BatchedMCTS interface over
libtakzero_hip.so (hand-written HIP for gfx950).  Importing `takzero_amd.api` requires the built
library; there is no CPU fallback."""
from . import weights  # noqa: F401
