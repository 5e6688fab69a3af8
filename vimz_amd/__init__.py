"""vimz_amd — MI355X-native accelerator for the Nova folding hot path of VIMz.

The compute path is the HIP shared library behind include/vimz_hip.h; this package only holds the
ctypes binding and the host-side mirror of the reference's folding interface.
"""
__all__ = ["hip", "image_editor"]
