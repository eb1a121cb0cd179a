from .device_loader import DeviceSRLoader  # noqa: F401,E402
