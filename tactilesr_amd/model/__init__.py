from .tactileSR_model import TactileSR, MSRB, ResBlock  # noqa: F401
