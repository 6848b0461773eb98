"""MI355X-native render path behind the path_tracer_c C API.

The product is the shared library ``lib/libptgpu.so`` (hand-written HIP for
gfx950 + a C-ABI, sources under ``csrc/``); this Python package is the test
and bench harness around it: ``_abi`` (ctypes declarations), ``host`` (scene
construction / rendering helpers over the ABI) and ``tiles`` (row-tile
sharding + the one gather for multi-GPU runs).
"""
from . import _abi, host, tiles  # noqa: F401
from ._abi import PtgpuError, lib  # noqa: F401
from .host import (CONFIGS, DeviceScene, HostScene, make_params,  # noqa: F401
                   render_frame, trace_rays)

__all__ = ["lib", "PtgpuError", "HostScene", "DeviceScene", "CONFIGS",
           "make_params", "render_frame", "trace_rays", "tiles", "host"]
