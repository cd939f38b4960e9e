"""ludwig_amd -- host-side mirror of the reference's lattice-Boltzmann call
surface (lb_collide / lb_halo / lb_propagation of zazu29/ludwig) on top of
liblbmi.so, the MI355X-native C-ABI library in ludwig_amd/csrc.

There is no CPU path: importing is always possible, but creating an LB object
fails loudly when the HIP library or a gfx950 device is missing.
"""

from .lib import build, library, LbmiError  # noqa: F401
from .lb import LB, Hydro, Ring, x_schedule, model, io_metadata_write, io_metadata_write_fmt, io_single_metadata_write, io_metadata_write_file, io_filename, M10, BGK, TRT, EAGER, FUSED, INPLACE, FUSED_HALO, FUSED_SOA, HALO_FULL, HALO_REDUCED  # noqa: F401
from .decomp import SlabDecomposition, CartDecomposition  # noqa: F401
