from .device_loader import DeviceSRLoader  # noqa: F401,E402
from . import seqs_depth2tactile  # noqa: F401,E402
