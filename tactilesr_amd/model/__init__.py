from .tactileSR_model import TactileSR, TactileSRCNN, MSRB, ResBlock  # noqa: F401
