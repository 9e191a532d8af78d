"""Small helpers shared by the host-side mirror of the reference interface."""
import logging

from torch import optim


class NotInstantiatedError(Exception):
    """Raised by any victim method called before .I(...) (recad/utils.py:217-227)."""


class InstantiateFail(Exception):
    """Raised when the deferred constructor throws (recad/utils.py:207-208)."""


class VarDim:
    """Symbolic variable-length dimension for input_describe()/output_describe(); prints as
    ``comment[min~max]`` like the reference's (recad/utils.py:128-138)."""

    __slots__ = ("max", "min", "comment")

    def __init__(self, max=None, min=None, comment=""):
        self.max = "?" if not max else max
        self.min = "0" if not min else min
        self.comment = comment

    def __repr__(self):
        return "%s[%s~%s]" % (self.comment, self.min, self.max)

    __str__ = __repr__


def get_logger(name, level=None):
    logger = logging.getLogger(name)
    if not logger.handlers:
        h = logging.StreamHandler()
        h.setFormatter(logging.Formatter("%(asctime)s %(name)s %(levelname)s %(message)s", datefmt="%H:%M:%S"))
        logger.addHandler(h)
    logger.setLevel(level or logging.INFO)
    return logger


def pick_optim(which):
    """Optimizer class by name: 'adam' (any case) -> torch.optim.Adam, otherwise the torch.optim
    attribute of that exact name; unknown names raise ValueError (recad/utils.py:181-189)."""
    name = str(which)
    if name.lower() == "adam":
        return optim.Adam
    cls = getattr(optim, name, None)
    if cls is None:
        raise ValueError("optimizer not supported")
    return cls


def parse_args(args):
    if isinstance(args, str):
        return [s.strip() for s in args.split(",") if s.strip()]
    return list(args)


class NullProgress:
    def set_description(self, *a, **k):
        return None
