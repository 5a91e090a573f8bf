"""qo-100-tools_amd — MI355X-native IF-chain FIR filter/decimator (host-side Python mirror of include/if_fir.h).

The product is libif_fir.so (HIP kernels + C-ABI, see csrc/).  This package only binds it with ctypes for the
tests and bench.py; it contains no compute and no CPU fallback: importing `if_fir` without the built library, or
creating a filter without a HIP device, raises.

The directory name is not a Python identifier; load it with `__graft_entry__.load_pkg()` (registers it as
`qo100_tools_amd`).
"""
from . import if_fir  # noqa: F401
from . import channel_shard  # noqa: F401
from . import wb_detect  # noqa: F401
from . import rc_reg  # noqa: F401
