"""Drop-in `simulator` package backed by libbgs.so (HIP kernels for gfx950); see include/bgs.h and DESIGN.md.

Importing the package changes nothing in the process: in particular it does not touch GPU_MAX_HW_QUEUES (round 3 did,
for every HIP user of the process).  Pipelines deeper than the HIP runtime's 4 default hardware queues ask for more
themselves, when they are built (`simulator.pipeline.request_hardware_queues`)."""

__version__ = "0.0.6+mi355x.2"

from .game import _abi as _abi
