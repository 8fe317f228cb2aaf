"""fastgen_amd — MI355X-native few-step diffusion sampling path (EDM U-Net + DMD2/sCM-style student sampler).

Host-side mirror of the reference's interfaces for that one path; all arithmetic on the path runs in
libfastgen_amd.so (hand-written HIP for gfx950) behind the C ABI of include/fastgen_amd.h.
"""
__version__ = "0.1"
