# -*- coding:utf-8 -*-
"""Drop-in for `deeptables.models.config.ModelConfig` (deeptables/models/config.py:9-216): an
immutable record with the same 45 fields and defaults.  Pure host code — no compute."""
import collections
import copy
import os

from ..utils import consts
from . import deepnets

# field -> default, in the reference's positional order (config.py:10-55 / 59-141)
_DEFAULTS = collections.OrderedDict([
    ('name', 'conf-1'),
    ('nets', ['dnn_nets']),
    ('categorical_columns', 'auto'),
    ('exclude_columns', []),
    ('task', consts.TASK_AUTO),
    ('pos_label', None),
    ('metrics', ['accuracy']),
    ('auto_categorize', False),
    ('cat_exponent', 0.5),
    ('cat_remain_numeric', True),
    ('auto_encode_label', True),
    ('auto_imputation', True),
    ('auto_scale', False),
    ('auto_discrete', False),
    ('auto_discard_unique', True),
    ('apply_gbm_features', False),
    ('gbm_params', {}),
    ('gbm_feature_type', consts.GBM_FEATURE_TYPE_EMB),
    ('fixed_embedding_dim', True),
    ('embeddings_output_dim', 4),
    ('embeddings_initializer', 'uniform'),
    ('embeddings_regularizer', None),
    ('embeddings_activity_regularizer', None),
    ('dense_dropout', 0),
    ('embedding_dropout', 0.3),
    ('stacking_op', consts.STACKING_OP_ADD),
    ('output_use_bias', True),
    ('apply_class_weight', False),
    ('optimizer', 'auto'),
    ('loss', 'auto'),
    ('dnn_params', {'hidden_units': ((128, 0, False), (64, 0, False)), 'activation': 'relu'}),
    ('autoint_params', {'num_attention': 3, 'num_heads': 1, 'dropout_rate': 0, 'use_residual': True}),
    ('fgcnn_params', {'fg_filters': (14, 16), 'fg_heights': (7, 7), 'fg_pool_heights': (2, 2),
                      'fg_new_feat_filters': (2, 2)}),
    ('fibinet_params', {'senet_pooling_op': 'mean', 'senet_reduction_ratio': 3,
                        'bilinear_type': 'field_interaction'}),
    ('cross_params', {'num_cross_layer': 4}),
    ('pnn_params', {'outer_product_kernel_type': 'mat'}),
    ('afm_params', {'attention_factor': 4, 'dropout_rate': 0}),
    ('cin_params', {'cross_layer_size': (128, 128), 'activation': 'relu', 'use_residual': False,
                    'use_bias': False, 'direct': False, 'reduce_D': False}),
    ('home_dir', None),
    ('monitor_metric', None),
    ('earlystopping_patience', 1),
    ('earlystopping_mode', 'auto'),
    ('gpu_usage_strategy', consts.GPU_USAGE_STRATEGY_GROWTH),
    ('distribute_strategy', None),
    ('var_len_categorical_columns', None),
])


class ModelConfig(collections.namedtuple('ModelConfig', list(_DEFAULTS.keys()))):
    def __hash__(self):
        return self.name.__hash__()

    def __new__(cls, *args, **kwargs):
        names = list(_DEFAULTS.keys())
        if len(args) > len(names):
            raise TypeError(f'ModelConfig takes at most {len(names)} arguments')
        values = {k: copy.deepcopy(v) for k, v in _DEFAULTS.items()}
        for k, v in zip(names, args):
            values[k] = v
        for k, v in kwargs.items():
            if k not in values:
                raise TypeError(f"ModelConfig got an unexpected keyword argument '{k}'")
            values[k] = v

        vlc = values['var_len_categorical_columns']
        if vlc is not None and len(vlc) > 0:     # config.py:137-149
            for v in vlc:
                if not isinstance(v, (tuple, list)) or len(v) != 3:
                    raise ValueError("Var len column config should be a tuple 3.")
                _name = v[0]
                if values['exclude_columns'] is not None and _name in values['exclude_columns']:
                    raise ValueError(f"Var len column {_name} can not put in 'exclude_columns' ")
                cc = values['categorical_columns']
                if cc is not None and isinstance(cc, list) and _name in cc:
                    raise ValueError(f"Var len column {_name} can not put in 'categorical_columns' ")

        values['nets'] = deepnets.get_nets(values['nets'])      # config.py:151
        if values['home_dir'] is None and os.environ.get(consts.ENV_DEEPTABLES_HOME) is not None:
            values['home_dir'] = os.environ.get(consts.ENV_DEEPTABLES_HOME)
        return super().__new__(cls, **values)

    @property
    def first_metric_name(self):
        if self.metrics is None or len(self.metrics) <= 0:
            raise ValueError('`metrics` is none or empty.')
        first_metric = self.metrics[0]
        if isinstance(first_metric, str):
            return first_metric
        if hasattr(first_metric, 'name'):
            return first_metric.name
        if callable(first_metric):
            return first_metric.__name__
        raise ValueError('`metric` must be string or callable object.')
