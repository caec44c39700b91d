# -*- coding:utf-8 -*-
"""Thin drop-in for `deeptables.models.deeptable.DeepTable` (deeptables/models/deeptable.py:30-822):
`DeepTable(config).fit(X_df, y) -> (DeepModel, History)`, `predict`, `predict_proba`, `evaluate`,
`save`/`load`.  Orchestration only — cross-validation, ensembling, GBM features and the
hypernets toolbox are outside the accelerated hot path.  The default preprocessor is
`preprocessor.DefaultPreprocessor` (the TF/hypernets-free restatement of the reference's pipeline);
`SimplePreprocessor` below is a smaller impute + label-encode variant kept for callers that want ids with 0
reserved for unseen values."""
import os
import pickle

import numpy as np
import pandas as pd

from . import deepmodel
from .config import ModelConfig
from .metainfo import CategoricalColumn, ContinuousColumn
from .preprocessor import DefaultPreprocessor
from ..utils import consts


class SimplePreprocessor:
    """impute -> label-encode categoricals -> pass numerics through (float32)."""

    def __init__(self, config):
        self.config = config
        self.categorical_columns = None
        self.continuous_columns = None
        self.labels_ = None
        self.task_ = None
        self._cat_maps = {}
        self._num_fill = {}
        self._num_scale = {}
        self.pos_label = config.pos_label

    # -- y ----------------------------------------------------------------------------------------
    def _infer_task(self, y):
        if self.config.task != consts.TASK_AUTO:
            task = self.config.task
            labels = sorted(pd.unique(pd.Series(y))) if task != consts.TASK_REGRESSION else []
            return task, list(labels)
        ys = pd.Series(y)
        if ys.dtype.kind == 'f' and ys.nunique() > 20:
            return consts.TASK_REGRESSION, []
        labels = sorted(ys.dropna().unique())
        if len(labels) == 2:
            return consts.TASK_BINARY, list(labels)
        if len(labels) > 2 and (ys.dtype.kind in 'OUSb' or len(labels) <= 100):
            return consts.TASK_MULTICLASS, list(labels)
        return consts.TASK_REGRESSION, []

    def transform_y(self, y):
        if self.task_ == consts.TASK_REGRESSION or not self.config.auto_encode_label:
            return np.asarray(y, dtype=np.float32)
        mapping = {v: i for i, v in enumerate(self.labels_)}
        return pd.Series(y).map(mapping).values.astype(np.float32)

    def inverse_transform_y(self, y_indicator):
        return np.asarray(self.labels_)[np.asarray(y_indicator).astype(np.int64)]

    # -- X ----------------------------------------------------------------------------------------
    def fit_transform(self, X, y):
        X = X.copy()
        X.columns = [str(c) for c in X.columns]
        drop = [c for c in (self.config.exclude_columns or []) if c in X.columns]
        X = X.drop(columns=drop)
        self.task_, self.labels_ = self._infer_task(y)
        if self.pos_label is not None and self.task_ == consts.TASK_BINARY and self.pos_label in self.labels_:
            self.labels_ = [l for l in self.labels_ if l != self.pos_label] + [self.pos_label]
        cc = self.config.categorical_columns
        cat_names = []
        for c in X.columns:
            kind = X[c].dtype.kind
            if (isinstance(cc, (list, tuple)) and c in cc) or kind in 'OUSb' or str(X[c].dtype) == 'category':
                cat_names.append(c)
            elif cc == 'auto' and self.config.auto_categorize and kind in 'iu' and \
                    X[c].nunique() < len(X) ** self.config.cat_exponent:
                cat_names.append(c)
        num_names = [c for c in X.columns if c not in cat_names]
        self.cat_names_, self.num_names_ = cat_names, num_names
        for c in cat_names:
            values = pd.unique(X[c].dropna())
            self._cat_maps[c] = {v: i + 1 for i, v in enumerate(sorted(values, key=lambda t: str(t)))}
        for c in num_names:
            col = pd.to_numeric(X[c], errors='coerce')
            self._num_fill[c] = float(col.mean()) if col.notna().any() else 0.0
            if self.config.auto_scale:
                sd = float(col.std()) or 1.0
                self._num_scale[c] = (float(col.mean()), sd if sd > 0 else 1.0)
        dim = self.config.embeddings_output_dim if self.config.fixed_embedding_dim else 0
        self.categorical_columns = [CategoricalColumn(c, len(self._cat_maps[c]) + 2, dim if dim > 0 else 0)
                                    for c in cat_names]
        self.continuous_columns = [ContinuousColumn(consts.INPUT_PREFIX_NUM + 'all', num_names)] if num_names else []
        return self.transform_X(X, _prepared=True), self.transform_y(y)

    def transform_X(self, X, _prepared=False):
        if not _prepared:
            X = X.copy()
            X.columns = [str(c) for c in X.columns]
        out = pd.DataFrame(index=X.index)
        for c in self.cat_names_:
            out[c] = X[c].map(self._cat_maps[c]).fillna(0).astype(np.int64)     # unseen / NaN -> 0
        for c in self.num_names_:
            col = pd.to_numeric(X[c], errors='coerce').fillna(self._num_fill[c]).astype(np.float32)
            if c in self._num_scale:
                mu, sd = self._num_scale[c]
                col = (col - mu) / sd
            out[c] = col
        return out

    def get_categorical_columns(self):
        return [c.name for c in self.categorical_columns]

    def get_continuous_columns(self):
        return list(self.num_names_)


class DeepTable:
    """`DeepTable(config=ModelConfig(nets=deepnets.DeepFM)).fit(df, y)` — deeptable.py:30."""

    def __init__(self, config=None, preprocessor=None):
        self.config = config if config is not None else ModelConfig()
        self.nets = self.config.nets
        self.preprocessor = preprocessor if preprocessor is not None else DefaultPreprocessor(self.config)
        self.model = None
        self._models = {}

    @property
    def task(self):
        return self.preprocessor.task_

    @property
    def num_classes(self):
        return len(self.preprocessor.labels_)

    @property
    def classes_(self):
        return self.preprocessor.labels_

    def fit(self, X=None, y=None, batch_size=128, epochs=1, verbose=1, callbacks=None, validation_split=0.2,
            validation_data=None, shuffle=True, class_weight=None, sample_weight=None, initial_epoch=0,
            steps_per_epoch=None, validation_steps=None, validation_freq=1, max_queue_size=10, workers=1,
            use_multiprocessing=False):
        if X is None or len(X) == 0:
            raise ValueError('X can not be empty.')
        X_t, y_t = self.preprocessor.fit_transform(X, y)
        if validation_data is not None:
            validation_data = (self.preprocessor.transform_X(validation_data[0]),
                               self.preprocessor.transform_y(validation_data[1]))
        if len(self.preprocessor.categorical_columns) == 0 and len(self.preprocessor.continuous_columns) == 0:
            raise ValueError('No valid input columns.')
        model = deepmodel.DeepModel(self.task, self.num_classes, self.config,
                                    self.preprocessor.categorical_columns, self.preprocessor.continuous_columns,
                                    var_categorical_len_columns=getattr(self.preprocessor,
                                                                        'var_len_categorical_columns', None))
        if class_weight is None and self.config.apply_class_weight and self.task != consts.TASK_REGRESSION:
            class_weight = self.get_class_weight(y_t)               # reference deeptable.py:354-355
        history = model.fit(X_t, y_t, batch_size=batch_size, epochs=epochs, verbose=verbose, callbacks=callbacks,
                            validation_split=validation_split, validation_data=validation_data, shuffle=shuffle,
                            class_weight=class_weight, sample_weight=sample_weight,
                            initial_epoch=initial_epoch, steps_per_epoch=steps_per_epoch,
                            validation_steps=validation_steps, validation_freq=validation_freq)
        self.model = model
        self._models['dt-1'] = model
        return model, history

    def get_class_weight(self, y):
        """'balanced' class weights n / (classes * count) over the encoded labels (reference deeptable.py:651-668:
        sklearn compute_class_weight('balanced'))."""
        yl = np.asarray(y)
        yl = yl.argmax(-1) if yl.ndim > 1 and yl.shape[-1] > 1 else yl.reshape(-1).astype(np.int64)
        n_cls = len(self.classes_) if self.classes_ is not None else int(yl.max()) + 1
        counts = np.bincount(yl, minlength=n_cls).astype(np.float64)
        return {k: float(len(yl) / (n_cls * c)) if c > 0 else 0.0 for k, c in enumerate(counts)}

    def _require_model(self):
        if self.model is None:
            raise ValueError('DeepTable is not fitted yet.')

    def predict_proba(self, X, batch_size=128, verbose=0, **kwargs):
        self._require_model()
        proba = self.model.predict(self.preprocessor.transform_X(X), batch_size=batch_size)
        if self.task == consts.TASK_BINARY and proba.shape[-1] == 1:   # fix_binary_predict_proba_result
            proba = np.hstack([1.0 - proba, proba])
        return proba

    def predict(self, X, encode_to_label=True, batch_size=128, verbose=0, **kwargs):
        proba = self.predict_proba(X, batch_size=batch_size)
        if self.task == consts.TASK_REGRESSION:
            return proba.reshape(-1) if proba.shape[-1] == 1 else proba
        idx = proba.argmax(axis=-1)
        if encode_to_label and self.config.auto_encode_label:
            return self.preprocessor.inverse_transform_y(idx)
        return idx

    def evaluate(self, X_test, y_test, batch_size=256, verbose=0, **kwargs):
        self._require_model()
        return self.model.evaluate(self.preprocessor.transform_X(X_test), self.preprocessor.transform_y(y_test),
                                   batch_size=batch_size)

    def apply(self, X, output_layers, concat_outputs=False, batch_size=128, verbose=0, transformer=None):
        self._require_model()
        return self.model.apply(self.preprocessor.transform_X(X), output_layers, concat_outputs, batch_size,
                                verbose, transformer)

    def save(self, filepath, deepmodel_basename=None):
        self._require_model()
        os.makedirs(filepath, exist_ok=True)
        name = deepmodel_basename or 'dt-1'
        self.model.save(os.path.join(filepath, f'{name}.safetensors'))
        cfg = self.config._replace(distribute_strategy=None)      # strategy is stripped on pickle (deeptable.py:756-771)
        with open(os.path.join(filepath, 'dt.pkl'), 'wb') as f:
            pickle.dump({'config': cfg, 'preprocessor': self.preprocessor, 'model_file': f'{name}.safetensors'}, f,
                        protocol=4)

    @staticmethod
    def load(filepath):
        with open(os.path.join(filepath, 'dt.pkl'), 'rb') as f:
            blob = pickle.load(f)
        dt = DeepTable(config=blob['config'], preprocessor=blob['preprocessor'])
        dt.model = deepmodel.DeepModel(dt.task, dt.num_classes, dt.config, dt.preprocessor.categorical_columns,
                                       dt.preprocessor.continuous_columns,
                                       model_file=os.path.join(filepath, blob['model_file']),
                                       var_categorical_len_columns=getattr(dt.preprocessor,
                                                                           'var_len_categorical_columns', None))
        return dt
