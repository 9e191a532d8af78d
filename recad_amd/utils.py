"""Small helpers shared by the host-side mirror of the reference interface."""
import logging

from torch import optim


class NotInstantiatedError(Exception):
    """Raised by any victim method called before .I(...) (recad/utils.py:217-227)."""


class InstantiateFail(Exception):
    """Raised when the deferred constructor throws (recad/utils.py:207-208)."""


class VarDim:
    """Symbolic variable dimension used by input_describe()/output_describe() (recad/utils.py:128-138)."""

    def __init__(self, max=None, min=None, comment=""):
        self.max = max or "?"
        self.min = min or "0"
        self.comment = comment

    def __repr__(self):
        return f"{self.comment}[{self.min}~{self.max}]"


def get_logger(name, level=None):
    logger = logging.getLogger(name)
    if not logger.handlers:
        h = logging.StreamHandler()
        h.setFormatter(logging.Formatter("%(asctime)s %(name)s %(levelname)s %(message)s", datefmt="%H:%M:%S"))
        logger.addHandler(h)
    logger.setLevel(level or logging.INFO)
    return logger


def pick_optim(which):
    """recad/utils.py:181-189: 'adam' -> torch.optim.Adam, else any torch.optim class by name."""
    if which.lower() == "adam":
        return optim.Adam
    if hasattr(optim, which):
        return getattr(optim, which)
    raise ValueError("optimizer not supported")


def parse_args(args):
    if isinstance(args, str):
        return [s.strip() for s in args.split(",") if s.strip()]
    return list(args)


class NullProgress:
    def set_description(self, *a, **k):
        return None
