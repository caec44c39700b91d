# -*- coding:utf-8 -*-
"""deeptables_amd — MI355X-native engine for the DeepTables `models.layers` hot path.

Drop-in surface (same names / signatures as the reference): `deeptables_amd.models.layers`,
`.models.deepnets`, `.models.config.ModelConfig`, `.models.deepmodel.DeepModel`,
`.models.deeptable.DeepTable`.  Compute goes through the C-ABI in include/dt_hip.h
(libdt_hip.so, hand-written gfx950 HIP kernels); there is no CPU fallback for it.
"""
__version__ = '0.1.0'
