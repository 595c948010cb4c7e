"""tiny-cuda-nn compatible Python surface on top of libnerfvo_hip.so (MI355X / gfx950).

Mirrors ``import tinycudann as tcnn`` as nerfstudio's fields use it (inner
boundary): ``tcnn.Encoding``, ``tcnn.Network``, ``tcnn.NetworkWithInputEncoding`` plus the module
level helpers.  The arithmetic is in hand-written HIP kernels behind the C-ABI; this file is glue.
"""
from .modules import (  # noqa: F401
    Encoding,
    Module,
    Network,
    NetworkWithInputEncoding,
    batch_size_granularity,
    default_loss_scale,
    free_temporary_memory,
    preferred_precision,
    supports_jit_fusion,
)
