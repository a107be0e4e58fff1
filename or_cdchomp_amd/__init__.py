"""or_cdchomp_amd -- MI355X-native CHOMP hot path behind the orcdchomp command surface.

Package contents (only what the path needs):
  csrc/        HIP kernels (gfx950), host layer and the C ABI (include/orcdchomp_amd.h)
  module.py    Python mirror of the module object (SendCommand + batch API over ctypes)
  bindings.py  keyword front end emitting the reference's command strings (bind/runchomp)
  robots.py    robot descriptions (WAM-like arm with the reference sphere table, 30-dof tree)
  scenes.py    synthetic scenes used by the configs
"""
from .module import Module  # noqa: F401
from . import bindings, robots, scenes  # noqa: F401
