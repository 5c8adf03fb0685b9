"""Drop-in `simulator` package backed by libbgs.so (HIP kernels for gfx950); see include/bgs.h and DESIGN.md."""

__version__ = "0.0.6+mi355x.1"
