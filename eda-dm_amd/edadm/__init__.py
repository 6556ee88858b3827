"""edadm — host side of the MI355X-native EDA-DM hot path.

`edadm.lib` is the ctypes binding of libedadm.so (the C ABI of include/edadm.h);
`edadm.ops` wraps each entry point for torch device tensors (pointers + current HIP stream);
`edadm.engine` is the frozen int8 UNet executor used by `qdiff.QuantModel` at sampling time.
There is no CPU fallback: if the shared library is missing every op raises.
"""
from . import lib  # noqa: F401
