"""Model registry + NCSN++/DDPM++ (the `models` package of the reference, hot-path subset)."""
from . import utils  # noqa: F401
from . import ncsnpp  # noqa: F401  (registers 'ncsnpp')
from . import ema  # noqa: F401
