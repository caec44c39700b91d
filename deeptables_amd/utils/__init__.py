from . import consts  # noqa: F401
