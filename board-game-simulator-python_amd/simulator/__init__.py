"""Drop-in `simulator` package backed by libbgs.so (HIP kernels for gfx950); see include/bgs.h and DESIGN.md."""

__version__ = "0.0.6+mi355x.1"

from .game import _abi as _abi

# more than the HIP runtime's default 4 hardware queues, asked for while the runtime has not read the variable yet
# (simulator.game._abi._more_hardware_queues; pipelines with more than 4 batches in flight need them)
_abi._more_hardware_queues()
