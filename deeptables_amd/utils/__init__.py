from . import consts, counter  # noqa: F401
