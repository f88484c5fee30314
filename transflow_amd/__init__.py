"""transflow_amd -- MI355X (gfx950) backend for transflow's per-frame hot loop.

Farnebäck dense optical flow + the compositor remap, as hand-written HIP kernels
behind a C ABI (include/tfhip.h), exposed to Python through the reference's own
FlowSource / Compositor surface.  Importing the package
makes no GPU call.
"""
__version__ = "0.1.0"
