# -*- coding:utf-8 -*-
"""TF-free, hypernets-free stand-in for `deeptables.models.preprocessor` (SURVEY §8 f4;
deeptables/models/preprocessor.py:26-515): same class names, config switches, step order and column metadata, on
pandas + scikit-learn only, so `DeepTable.fit(df, y)` runs on raw frames.

Step order of `fit_transform` (preprocessor.py:165-204): y label-encode -> feature typing (`_prepare_features`)
-> imputation -> min-max scale -> categorical label encoding -> discretisation -> var-len encoding.
`X_transformers` keeps the fitted steps in that order and `transform_X` replays them (preprocessor.py:249-260).

Where the reference defers to `hypernets.tabular.sklearn_ex` (absent here) the behaviour assumed is written next
to the transformer: it only has to be self-consistent between fit and transform — the embedding gather sees ids in
`[0, vocabulary_size)` either way (vocabulary_size = nunique + 2, preprocessor.py:333).
GBM leaf features (`apply_gbm_features`, needs lightgbm) and the fit cache are not provided.
"""
import collections
import copy

import numpy as np
import pandas as pd

from .config import ModelConfig
from .metainfo import CategoricalColumn, ContinuousColumn, VarLenCategoricalColumn
from ..utils import consts


# ---------------------------------------------------------------------------------------------
# transformers (stand-ins for hypernets.tabular.sklearn_ex)
# ---------------------------------------------------------------------------------------------
class PassThroughEstimator:
    def fit(self, X, y=None):
        return self

    def transform(self, X):
        return X

    def fit_transform(self, X, y=None):
        return X


class CategorizeEncoder:
    """Low-cardinality numeric columns become categorical: with `remain_numeric` a string copy `<c>_cat` is added
    and the numeric column stays (preprocessor.py:320-327), otherwise the column itself is converted."""

    def __init__(self, columns, remain_numeric=True):
        self.columns = list(columns)
        self.remain_numeric = remain_numeric
        self.new_columns = []

    def fit(self, X, y=None):
        self.new_columns = []
        if self.remain_numeric:
            for c in self.columns:
                self.new_columns.append((c + '_cat', 'str', int(X[c].nunique())))
        return self

    def transform(self, X):
        for c in self.columns:
            target = c + '_cat' if self.remain_numeric else c
            X[target] = X[c].astype(str)
        return X

    def fit_transform(self, X, y=None):
        return self.fit(X).transform(X)


class ColumnImputer:
    """SimpleImputer per column group: mean for continuous, '' / 0 constants for object / numeric categoricals
    (preprocessor.py:337-381)."""

    def __init__(self, continuous, obj_cats, num_cats):
        self.continuous, self.obj_cats, self.num_cats = list(continuous), list(obj_cats), list(num_cats)
        self.means_ = {}

    def fit(self, X, y=None):
        for c in self.continuous:
            col = pd.to_numeric(X[c], errors='coerce')
            self.means_[c] = float(col.mean()) if col.notna().any() else 0.0
        return self

    def transform(self, X):
        for c in self.continuous:
            X[c] = pd.to_numeric(X[c], errors='coerce').fillna(self.means_[c])
        for c in self.obj_cats:
            X[c] = X[c].astype(object).where(X[c].notna(), '')
        for c in self.num_cats:
            X[c] = X[c].fillna(0)
        return X

    def fit_transform(self, X, y=None):
        return self.fit(X).transform(X)


class MinMaxScalerTransformer:
    def __init__(self, columns):
        self.columns = list(columns)
        self.min_, self.range_ = {}, {}

    def fit(self, X, y=None):
        for c in self.columns:
            lo, hi = float(X[c].min()), float(X[c].max())
            self.min_[c], self.range_[c] = lo, (hi - lo) if hi > lo else 1.0
        return self

    def transform(self, X):
        for c in self.columns:
            X[c] = (X[c].astype('float64') - self.min_[c]) / self.range_[c]
        return X

    def fit_transform(self, X, y=None):
        return self.fit(X).transform(X)


class MultiLabelEncoder:
    """One safe label encoder per column: known values -> 0..n-1 (sorted by their string form), values unseen at
    fit time -> n.  n+1 <= nunique+2 = the column's vocabulary_size, so every id has an embedding row."""

    def __init__(self, columns):
        self.columns = list(columns)
        self.maps_ = {}

    def fit(self, X, y=None):
        for c in self.columns:
            values = sorted(pd.unique(X[c].astype(str)))
            self.maps_[c] = {v: i for i, v in enumerate(values)}
        return self

    def transform(self, X):
        for c in self.columns:
            m = self.maps_[c]
            X[c] = X[c].astype(str).map(m).fillna(len(m)).astype('int64')
        return X

    def fit_transform(self, X, y=None):
        return self.fit(X).transform(X)


class MultiKBinsDiscretizer:
    """Quantile bins per continuous column into a new ordinal column `<c>_discrete`
    (bins = min(nunique, 10) assumed; the reference takes the count from hypernets).  new_columns holds
    (name, new_name, bins) as preprocessor.py:410 expects."""

    def __init__(self, columns, max_bins=10):
        self.columns = list(columns)
        self.max_bins = max_bins
        self.new_columns = []
        self.edges_ = {}

    def fit(self, X, y=None):
        self.new_columns = []
        for c in self.columns:
            col = X[c].astype('float64')
            bins = int(max(2, min(self.max_bins, col.nunique())))
            edges = np.unique(np.quantile(col, np.linspace(0, 1, bins + 1)))
            self.edges_[c] = edges[1:-1]
            self.new_columns.append((c, c + '_discrete', len(edges) - 1 if len(edges) > 1 else 1))
        return self

    def transform(self, X):
        for c, new, _bins in self.new_columns:
            X[new] = np.searchsorted(self.edges_[c], X[c].astype('float64').values, side='right').astype('int64')
        return X

    def fit_transform(self, X, y=None):
        return self.fit(X).transform(X)


class MultiVarLenFeatureEncoder:
    """'a|b|c' strings -> zero-padded id lists of the column's max length; id 0 = padding / unseen, tokens 1..n
    (config.var_len_categorical_columns entries are (name, sep, pooling), config.py:138-144)."""

    def __init__(self, features):
        self.features = [(f[0], f[1]) for f in features]
        self.maps_, self.max_length_ = {}, {}

    @staticmethod
    def _split(v, sep):
        if v is None or (isinstance(v, float) and np.isnan(v)) or v == '':
            return []
        return [t for t in str(v).split(sep) if t != '']

    def fit(self, X, y=None):
        for name, sep in self.features:
            toks = [self._split(v, sep) for v in X[name]]
            vocab = sorted({t for row in toks for t in row})
            self.maps_[name] = {t: i + 1 for i, t in enumerate(vocab)}
            self.max_length_[name] = max([len(r) for r in toks] + [1])
        return self

    def transform(self, X):
        for name, sep in self.features:
            m, L = self.maps_[name], self.max_length_[name]
            rows = []
            for v in X[name]:
                ids = [m.get(t, 0) for t in self._split(v, sep)][:L]
                rows.append(np.asarray(ids + [0] * (L - len(ids)), dtype=np.int64))
            X[name] = rows
        return X

    def fit_transform(self, X, y=None):
        return self.fit(X).transform(X)

    def vocabulary_size(self, name):
        return len(self.maps_[name]) + 1


class LabelEncoder:
    def fit(self, y):
        self.classes_ = np.asarray(sorted(pd.unique(pd.Series(np.asarray(y).ravel()))))
        self._map = {v: i for i, v in enumerate(self.classes_)}
        return self

    def transform(self, y):
        s = pd.Series(np.asarray(y).ravel()).map(self._map)
        if s.isna().any():
            raise ValueError('y contains previously unseen labels')
        return s.values.astype('int64')

    def fit_transform(self, y):
        return self.fit(y).transform(y)

    def inverse_transform(self, idx):
        return self.classes_[np.asarray(idx).astype('int64')]


def infer_task_type(y):
    """hypernets' toolbox rule as used by preprocessor.py:208-209: float targets with many values -> regression,
    two classes -> binary, otherwise multiclass."""
    ys = pd.Series(np.asarray(y).ravel())
    n = ys.nunique()
    if ys.dtype.kind == 'f' and n > 2 and not np.all(np.mod(ys.dropna(), 1) == 0):
        return consts.TASK_REGRESSION, []
    if ys.dtype.kind in 'fiu' and n > 100:
        return consts.TASK_REGRESSION, []
    labels = sorted(ys.dropna().unique())
    return (consts.TASK_BINARY if n == 2 else consts.TASK_MULTICLASS), labels


# ---------------------------------------------------------------------------------------------
# preprocessors
# ---------------------------------------------------------------------------------------------
class AbstractPreprocessor:
    def __init__(self, config: ModelConfig):
        self.config = config
        self.labels_ = None
        self.task_ = None

    @property
    def pos_label(self):
        if self.labels_ is not None and len(self.labels_) == 2:
            return self.labels_[1]
        return None

    @property
    def labels(self):
        return self.labels_

    @property
    def task(self):
        return self.task_

    def fit_transform(self, X, y, copy_data=True):
        raise NotImplementedError

    def transform_X(self, X, copy_data=True):
        raise NotImplementedError

    def transform_y(self, y, copy_data=True):
        raise NotImplementedError

    def transform(self, X, y, copy_data=True):
        raise NotImplementedError

    def inverse_transform_y(self, y_indicator):
        raise NotImplementedError

    def get_categorical_columns(self):
        raise NotImplementedError

    def get_continuous_columns(self):
        raise NotImplementedError


class DefaultPreprocessor(AbstractPreprocessor):
    def __init__(self, config: ModelConfig):
        super().__init__(config)
        self.reset()

    def reset(self):
        self.metainfo = None
        self.categorical_columns = None
        self.var_len_categorical_columns = None
        self.continuous_columns = None
        self.y_lable_encoder = None
        self.X_transformers = collections.OrderedDict()

    # -- validation / preparation (preprocessor.py:116-157) ---------------------------------------
    def _validate_fit_transform(self, X, y):
        if X is None:
            raise ValueError('X cannot be none.')
        if y is None:
            raise ValueError('y cannot be none.')
        if len(X.shape) != 2:
            raise ValueError('X must be a 2D datasets.')
        if X.shape[0] != np.shape(y)[0]:
            raise ValueError(f'The number of samples of X and y must be the same. X.shape:{X.shape}, '
                             f'y.shape{np.shape(y)}')
        if pd.DataFrame(y).isnull().sum().sum() > 0:
            raise ValueError('Missing values in y.')

    def _prepare_X(self, X):
        if not isinstance(X, pd.DataFrame):
            X = pd.DataFrame(X)
        if len(set(X.columns)) != len(list(X.columns)):
            cols = [item for item, count in collections.Counter(X.columns).items() if count > 1]
            raise ValueError(f'Columns with duplicate names in X: {cols}')
        if X.columns.dtype != 'object':
            X.columns = ['x_' + str(c) for c in X.columns]
        return X

    # -- fit ----------------------------------------------------------------------------------------
    def fit_transform(self, X, y, copy_data=True):
        self.reset()
        self._validate_fit_transform(X, y)
        X = copy.deepcopy(X)
        y = copy.deepcopy(y)
        y = self.fit_transform_y(y)
        X = self._prepare_X(X)
        X = self._prepare_features(X)
        if self.config.auto_imputation:
            X = self._imputation(X)
        if self.config.auto_scale:
            X = self._standard_scale(X)
        if self.config.auto_encode_label:
            X = self._categorical_encoding(X)
        if self.config.auto_discrete:
            X = self._discretization(X)
        if self.config.apply_gbm_features:
            raise NotImplementedError('apply_gbm_features needs lightgbm, which this environment does not have')
        var_len = self.config.var_len_categorical_columns
        if var_len is not None and len(var_len) > 0:
            X = self._var_len_encoder(X, var_len)
        self.X_transformers['last'] = PassThroughEstimator()
        cont_cols = self.get_continuous_columns()
        if len(cont_cols) > 0:
            X[cont_cols] = X[cont_cols].astype('float')
        return X, y

    def fit_transform_y(self, y):
        if self.config.task == consts.TASK_AUTO:
            self.task_, self.labels_ = infer_task_type(y)
        else:
            self.task_ = self.config.task
        if self.task_ in [consts.TASK_BINARY, consts.TASK_MULTICLASS]:
            self.y_lable_encoder = LabelEncoder()
            y = self.y_lable_encoder.fit_transform(y)
            self.labels_ = self.y_lable_encoder.classes_
        elif self.task_ == consts.TASK_MULTILABEL:
            self.labels_ = list(range(np.shape(y)[-1]))
        else:
            self.labels_ = []
            y = np.asarray(y)
        return y

    # -- transform ------------------------------------------------------------------------------------
    def transform(self, X, y, copy_data=True):
        return self.transform_X(X, copy_data), self.transform_y(y, copy_data)

    def transform_y(self, y, copy_data=True):
        if self.y_lable_encoder is not None:
            return self.y_lable_encoder.transform(y)
        return np.asarray(y)

    def transform_X(self, X, copy_data=True):
        if copy_data:
            X = copy.deepcopy(X)
        X = self._prepare_X(X)
        for step in self.X_transformers.values():
            X = step.transform(X)
        cont_cols = self.get_continuous_columns()
        if len(cont_cols) > 0:
            X[cont_cols] = X[cont_cols].astype('float')
        return X

    def inverse_transform_y(self, y_indicator):
        if self.y_lable_encoder is not None:
            return self.y_lable_encoder.inverse_transform(y_indicator)
        return y_indicator

    # -- steps ----------------------------------------------------------------------------------------
    def _prepare_features(self, X):
        """Column typing, preprocessor.py:267-336."""
        num_vars, convert2cat_vars, cat_vars = [], [], []
        if self.config.cat_exponent >= 1:
            raise ValueError(f'"cat_exponent" must be less than 1, not {self.config.cat_exponent} .')
        var_len = self.config.var_len_categorical_columns
        var_len_sep = {}
        if var_len is not None and len(var_len) > 0:
            for v in var_len:
                if not isinstance(v, (tuple, list)) or len(v) != 3:
                    raise ValueError('Var len column config should be a tuple 3.')
                var_len_sep[v[0]] = v[1]
        unique_upper_limit = round(X.shape[0] ** self.config.cat_exponent)
        for c in X.columns:
            nunique = X[c].nunique()
            dtype = str(X[c].dtype)
            if nunique <= 1 and self.config.auto_discard_unique:
                continue
            if c in (self.config.exclude_columns or []):
                continue
            if c in var_len_sep:
                self._append_var_len_categorical_col(c, nunique, var_len_sep[c])
                continue
            cc = self.config.categorical_columns
            if cc is not None and isinstance(cc, list):
                if c in cc:
                    cat_vars.append((c, dtype, nunique))
                elif np.issubdtype(X[c].dtype, np.number):
                    num_vars.append((c, dtype, nunique))
            else:
                if dtype in ('object', 'category', 'bool') or dtype.startswith('str'):
                    cat_vars.append((c, dtype, nunique))
                elif self.config.auto_categorize and nunique < unique_upper_limit:
                    convert2cat_vars.append((c, dtype, nunique))
                else:
                    num_vars.append((c, dtype, nunique))
        if len(convert2cat_vars) > 0:
            ce = CategorizeEncoder([c for c, d, n in convert2cat_vars], self.config.cat_remain_numeric)
            X = ce.fit_transform(X)
            self.X_transformers['categorize'] = ce
            if self.config.cat_remain_numeric:
                cat_vars = cat_vars + ce.new_columns
                num_vars = num_vars + convert2cat_vars
            else:
                cat_vars = cat_vars + convert2cat_vars
        self._append_categorical_cols([(c[0], c[2] + 2) for c in cat_vars])
        self._append_continuous_cols([c[0] for c in num_vars], consts.INPUT_PREFIX_NUM + 'all')
        return X

    def _imputation(self, X):
        obj_cats, num_cats = [], []
        for c in self.get_categorical_columns() + self.get_var_len_categorical_columns():
            dtype = str(X[c].dtype)
            (obj_cats if dtype.startswith('obj') or dtype.startswith('str') or dtype == 'category'
             else num_cats).append(c)
        imp = ColumnImputer(self.get_continuous_columns(), obj_cats, num_cats)
        X = imp.fit_transform(X)
        self.X_transformers['imputation'] = imp
        return X

    def _categorical_encoding(self, X):
        mle = MultiLabelEncoder(self.get_categorical_columns())
        X = mle.fit_transform(X)
        self.X_transformers['label_encoder'] = mle
        return X

    def _standard_scale(self, X):
        ss = MinMaxScalerTransformer(self.get_continuous_columns())
        X = ss.fit_transform(X)
        self.X_transformers['standard_scale'] = ss
        return X

    def _discretization(self, X):
        mkbd = MultiKBinsDiscretizer(self.get_continuous_columns())
        X = mkbd.fit_transform(X)
        self._append_categorical_cols([(new_name, bins + 1) for name, new_name, bins in mkbd.new_columns])
        self.X_transformers['discreter'] = mkbd
        return X

    def _var_len_encoder(self, X, var_len_categorical_columns):
        transformer = MultiVarLenFeatureEncoder(var_len_categorical_columns)
        X = transformer.fit_transform(X)
        updated = []
        for c in self.var_len_categorical_columns:
            # the column was sized from nunique of the raw strings; the table must cover the TOKEN vocabulary
            vc = VarLenCategoricalColumn(c.name, max(c.vocabulary_size, transformer.vocabulary_size(c.name)),
                                         c.embeddings_output_dim, sep=c.sep)
            vc.max_elements_length = transformer.max_length_[c.name]
            updated.append(vc)
        self.var_len_categorical_columns = updated
        self.X_transformers['var_len_encoder'] = transformer
        return X

    # -- column metadata (preprocessor.py:452-515) ----------------------------------------------------
    def _embedding_dim(self, voc_size):
        if self.config.fixed_embedding_dim:
            d = self.config.embeddings_output_dim if self.config.embeddings_output_dim > 0 \
                else consts.EMBEDDING_OUT_DIM_DEFAULT
            return d
        return min(4 * int(pow(voc_size, 0.25)), 20)

    def _append_var_len_categorical_col(self, name, voc_size, sep):
        if self.var_len_categorical_columns is None:
            self.var_len_categorical_columns = []
        self.var_len_categorical_columns.append(
            VarLenCategoricalColumn(name, voc_size, self._embedding_dim(voc_size), sep=sep))

    def _append_categorical_cols(self, cols):
        if self.categorical_columns is None:
            self.categorical_columns = []
        if cols is not None and len(cols) > 0:
            self.categorical_columns = self.categorical_columns + \
                [CategoricalColumn(name, voc_size, self._embedding_dim(voc_size)) for name, voc_size in cols]

    def _append_continuous_cols(self, cols, input_name):
        if self.continuous_columns is None:
            self.continuous_columns = []
        if cols is not None and len(cols) > 0:
            self.continuous_columns = self.continuous_columns + \
                [ContinuousColumn(name=input_name, column_names=[c for c in cols])]

    def get_categorical_columns(self):
        return [c.name for c in self.categorical_columns]

    def get_var_len_categorical_columns(self):
        if self.var_len_categorical_columns is not None:
            return [c.name for c in self.var_len_categorical_columns]
        return []

    def get_continuous_columns(self):
        cont_vars = []
        for c in self.continuous_columns:
            cont_vars = cont_vars + c.column_names
        return cont_vars
