from . import dsutils  # noqa: F401
