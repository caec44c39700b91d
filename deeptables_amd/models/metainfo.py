# -*- coding:utf-8 -*-
"""Column metadata of the drop-in API — same fields/defaults as deeptables/models/metainfo.py:33-86."""
import collections

from ..utils import consts


class CategoricalColumn(collections.namedtuple('CategoricalColumn',
                                               ['name', 'vocabulary_size', 'embeddings_output_dim', 'dtype',
                                                'input_name'])):
    def __hash__(self):
        return self.name.__hash__()

    def __new__(cls, name, vocabulary_size, embeddings_output_dim=10, dtype='int32', input_name=None):
        if input_name is None:
            input_name = consts.INPUT_PREFIX_CAT + name
        if embeddings_output_dim == 0:   # metainfo.py:46-47: auto size = vocab ** 0.25
            embeddings_output_dim = int(round(vocabulary_size ** 0.25))
        return super().__new__(cls, name, vocabulary_size, embeddings_output_dim, dtype, input_name)


class VarLenCategoricalColumn(collections.namedtuple('VarLenCategoricalColumn',
                                                     ['name', 'vocabulary_size', 'embeddings_output_dim', 'dtype',
                                                      'input_name', 'sep'])):
    def __hash__(self):
        return self.name.__hash__()

    def __new__(cls, name, vocabulary_size, embeddings_output_dim=10, dtype='int32', input_name=None, sep='|'):
        if input_name is None:
            input_name = consts.INPUT_PREFIX_CAT + name
        if embeddings_output_dim == 0:
            embeddings_output_dim = int(round(vocabulary_size ** 0.25))
        return super().__new__(cls, name, vocabulary_size, embeddings_output_dim, dtype, input_name, sep)


class ContinuousColumn(collections.namedtuple('ContinuousColumn',
                                              ['name', 'column_names', 'input_dim', 'dtype', 'input_name'])):
    def __hash__(self):
        return self.name.__hash__()

    def __new__(cls, name, column_names, input_dim=0, dtype='float32', input_name=None):
        input_dim = len(column_names)    # metainfo.py:85
        return super().__new__(cls, name, column_names, input_dim, dtype, input_name)
