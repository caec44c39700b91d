# -*- coding:utf-8 -*-
from .config import ModelConfig  # noqa: F401
from .deeptable import DeepTable  # noqa: F401
from .deepmodel import DeepModel  # noqa: F401
