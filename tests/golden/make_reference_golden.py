# -*- coding:utf-8 -*-
"""Pins the oracle to the REFERENCE'S OWN CODE.

TensorFlow / Keras cannot be installed here, so the reference cannot run as it is.  But its hot path
(/root/reference/deeptables/models/layers.py, deepnets.py, deepmodel.py) is a composition of a few dozen primitive
tensor ops and stock Keras layers (tf.reduce_sum, tf.split, tf.concat, tf.matmul, tf.tensordot, tf.nn.conv1d,
tf.nn.softmax, Dense, BatchNormalization, Conv2D, ...), and the PRIMITIVES have unambiguous published semantics.
This script

  1. installs an in-process shim of exactly those primitives on torch float64 (modules `tensorflow`, `keras`, ... in
     sys.modules; nothing is written anywhere),
  2. imports the reference's layers.py, deepnets.py, config.py, metainfo.py, utils/consts.py, utils/counter.py and
     deepmodel.py UNMODIFIED from /root/reference (read-only) on top of the shim,
  3. runs the reference's own code on seeded inputs and weights: `build` / `call` of every layer and loss class of
     layers.py; the net functions of deepnets.py; and whole models through DeepModel.__build_model with the reference's
     ModelConfig defaults (the five BASELINE.json configurations, every preset, stacking / head variants),
  4. checks the oracle (oracle/reference_layers.py) against those outputs to 1e-12, and
  5. writes inputs, weights and the reference code's outputs to tests/golden/reference_code_*.npz.

What this pins: the op ORDER and every shape / axis / split / transpose / wiring decision of the reference's code — the
part a restatement can get wrong.  What it cannot pin: TensorFlow's own arithmetic inside a primitive (float32 rounding,
reduction order) and the stock Keras layers' defaults (BatchNormalization epsilon, initializers), which the shim states
the same way the oracle does.  tests/test_oracle_reference_code.py compares the oracle with the committed vectors on
every CPU run; this script only runs where /root/reference exists.

    python tests/golden/make_reference_golden.py          # regenerates the fixtures (deterministic)
"""
import contextlib
import importlib.util
import inspect
import json
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
DT = torch.float64


# ---------------------------------------------------------------------------------------------------------------
# 1. the shim: tensors are torch tensors; TensorShape-style accessors the reference uses are added to torch.Tensor
# ---------------------------------------------------------------------------------------------------------------
class _Shape(tuple):
    def as_list(self):
        return list(self)


def _install_tensor_accessors():
    torch.Tensor.get_shape = lambda self: _Shape(int(s) for s in self.shape)
    torch.Tensor.numpy_ = lambda self: self.detach().cpu().numpy()


def _t(x):
    if isinstance(x, (list, tuple)) and len(x) and isinstance(x[0], torch.Tensor):
        return torch.stack(list(x), 0)          # TF converts a list of tensors to one stacked tensor
    if isinstance(x, torch.Tensor):
        return x
    return torch.as_tensor(x, dtype=DT)


def _axis(a):
    return tuple(a) if isinstance(a, (list, tuple)) else a


def tf_reduce_sum(x, axis=None, keepdims=False, keep_dims=None):
    kd = keepdims if keep_dims is None else keep_dims
    return _t(x).sum() if axis is None else _t(x).sum(dim=_axis(axis), keepdim=kd)


def tf_reduce_mean(x, axis=None, keepdims=False):
    return _t(x).mean() if axis is None else _t(x).mean(dim=_axis(axis), keepdim=keepdims)


def tf_reduce_max(x, axis=None, keepdims=False):
    return _t(x).max() if axis is None else _t(x).amax(dim=_axis(axis), keepdim=keepdims)


def tf_split(value, num_or_size_splits, axis=0):
    value = _t(value)
    if isinstance(num_or_size_splits, int):
        assert value.shape[axis] % num_or_size_splits == 0
        return list(torch.split(value, value.shape[axis] // num_or_size_splits, dim=axis))
    return list(torch.split(value, list(num_or_size_splits), dim=axis))


def tf_matmul(a, b, transpose_a=False, transpose_b=False):
    a, b = _t(a), _t(b)
    if transpose_a:
        a = a.transpose(-1, -2)
    if transpose_b:
        b = b.transpose(-1, -2)
    return torch.matmul(a, b)


def tf_tensordot(a, b, axes):
    a, b = _t(a), _t(b)
    if isinstance(axes, int):
        return torch.tensordot(a, b, dims=axes)
    ax_a, ax_b = axes
    ax_a = [ax_a] if isinstance(ax_a, int) else list(ax_a)
    ax_b = [ax_b] if isinstance(ax_b, int) else list(ax_b)
    return torch.tensordot(a, b, dims=(ax_a, ax_b))


def tf_conv1d(input, filters, stride=1, padding='VALID'):
    """tf.nn.conv1d, NWC input [batch, width, in_ch], filters [k, in_ch, out_ch], VALID."""
    assert padding == 'VALID'
    x = _t(input).permute(0, 2, 1)                       # -> [batch, in_ch, width]
    w = _t(filters).permute(2, 1, 0)                     # -> [out_ch, in_ch, k]
    return torch.nn.functional.conv1d(x, w, stride=stride).permute(0, 2, 1)


def _make_tf():
    tf = types.ModuleType('tensorflow')
    tf.float32, tf.int32, tf.int64 = 'float32', 'int32', 'int64'
    tf.reduce_sum, tf.reduce_mean, tf.reduce_max = tf_reduce_sum, tf_reduce_mean, tf_reduce_max
    tf.expand_dims = lambda x, axis: _t(x).unsqueeze(axis)
    tf.split = tf_split
    tf.concat = lambda values, axis: torch.cat([_t(v) for v in values], dim=axis)
    tf.transpose = lambda a, perm=None: _t(a).permute(*perm) if perm is not None else _t(a).t()
    tf.tensordot = tf_tensordot
    tf.reshape = lambda x, shape: _t(x).reshape(*[int(s) for s in shape])
    tf.matmul = tf_matmul
    tf.multiply = lambda a, b: _t(a) * _t(b)
    tf.square = lambda x: _t(x) * _t(x)
    tf.sigmoid = lambda x: torch.sigmoid(_t(x))
    tf.zeros_like = lambda x: torch.zeros_like(_t(x))
    tf.ones_like = lambda x: torch.ones_like(_t(x))
    tf.shape = lambda x: _t(x).shape
    tf.constant = lambda v, dtype=None: torch.as_tensor(v, dtype=DT)
    tf.where = lambda c, a, b: torch.where(c, _t(a), _t(b))
    tf.equal = lambda a, b: _t(a) == _t(b)
    tf.greater = lambda a, b: _t(a) > _t(b)
    tf.greater_equal = lambda a, b: _t(a) >= _t(b)
    tf.less = lambda a, b: _t(a) < _t(b)
    tf.logical_and = lambda a, b: a & b
    tf.abs = lambda x: torch.abs(_t(x))
    tf.maximum = lambda a, b: torch.clamp(_t(a), min=b) if not isinstance(b, torch.Tensor) else torch.maximum(_t(a), b)
    tf.cast = lambda x, dtype=None: (x.to(DT) if isinstance(x, torch.Tensor) else torch.as_tensor(float(x), dtype=DT))
    tf.Variable = lambda v, trainable=True, **kw: torch.as_tensor(v, dtype=DT).clone()
    tf.identity = lambda x, name=None: x
    tf.control_dependencies = lambda deps: contextlib.nullcontext()

    def assign(var, value):
        var.copy_(value)
        return var
    tf.compat = types.SimpleNamespace(v1=types.SimpleNamespace(assign=assign))
    nn = types.ModuleType('tensorflow.nn')
    nn.softmax = lambda x, axis=-1: torch.softmax(_t(x), dim=axis)
    nn.relu = lambda x: torch.relu(_t(x))
    nn.conv1d = tf_conv1d
    nn.bias_add = lambda x, b: _t(x) + _t(b)
    nn.sigmoid = lambda x: torch.sigmoid(_t(x))
    nn.sigmoid_cross_entropy_with_logits = lambda labels=None, logits=None: \
        torch.nn.functional.binary_cross_entropy_with_logits(_t(logits), _t(labels), reduction='none')
    tf.nn = nn
    return tf


# ---- keras.layers.* the reference's layers instantiate -----------------------------------------------------------------
_RNG = None
_INIT_SCALE = 1.0       # run_model(init_scale=...): deep cross stacks saturate the sigmoid with +-0.5 weights


def _init(shape, kind):
    """deterministic weights (the VALUES do not matter for pinning the op sequence; zeros would hide terms)"""
    shape = tuple(int(s) for s in shape)
    return torch.as_tensor(_RNG.uniform(-0.5, 0.5, size=shape) * _INIT_SCALE, dtype=DT).requires_grad_(True)   # (the gradient fixtures)


def _act(name):
    if name is None or name == 'linear':
        return lambda t: t
    if callable(name):
        return name
    return {'relu': torch.relu, 'sigmoid': torch.sigmoid, 'tanh': torch.tanh,
            'softmax': lambda t: torch.softmax(t, -1)}[name]


_REGISTRY = []          # every shim layer in creation order (the net functions create theirs internally)


class Layer:
    def __init__(self, name=None, **kwargs):
        self.name = name
        self.built = False
        self._weights = []
        _REGISTRY.append(self)

    def add_weight(self, name=None, shape=None, initializer=None, dtype=None, trainable=True, regularizer=None,
                   constraint=None, **kw):
        w = _init(shape, initializer)
        self._weights.append((name, w))
        return w

    def build(self, input_shape):
        self.built = True

    def __call__(self, x, **kwargs):
        if not self.built:
            if isinstance(x, (list, tuple)):
                self.build([_Shape(int(s) for s in t.shape) for t in x])
            else:
                self.build(_Shape(int(s) for s in x.shape))
            self.built = True
        return self.call(x, **kwargs)

    def get_config(self):
        return {}

    def compute_output_shape(self, input_shape):
        return input_shape


class Dense(Layer):
    def __init__(self, units, activation=None, use_bias=True, kernel_initializer=None, **kw):
        super().__init__(**kw)
        self.units, self.activation, self.use_bias = units, activation, use_bias

    def build(self, input_shape):
        self.kernel = self.add_weight('kernel', (input_shape[-1], self.units))
        self.bias = self.add_weight('bias', (self.units,)) if self.use_bias else None

    def call(self, x):
        y = torch.matmul(x, self.kernel)
        if self.bias is not None:
            y = y + self.bias
        self.last_preact = y
        return _act(self.activation)(y)


class Dropout(Layer):                       # inference / rate 0: identity
    def __init__(self, rate=0.0, **kw):
        super().__init__(**kw)
        self.rate = rate

    def call(self, x, training=None):
        assert not self.rate, 'the fixtures use dropout rate 0'
        return x


SpatialDropout1D = Dropout


class BatchNormalization(Layer):
    """keras.layers.BatchNormalization() in TRAINING mode: batch statistics over all axes but the last, biased variance,
    epsilon 1e-3, gamma ones / beta zeros at build (they are replaced by seeded values so the affine part is visible)."""

    def __init__(self, epsilon=1e-3, momentum=0.99, **kw):
        super().__init__(**kw)
        self.epsilon = epsilon

    def build(self, input_shape):
        self.gamma = self.add_weight('gamma', (input_shape[-1],)) + 1.0
        self.beta = self.add_weight('beta', (input_shape[-1],))

    def call(self, x, training=None):
        red = tuple(range(x.dim() - 1))
        mean = x.mean(dim=red, keepdim=True)
        var = ((x - mean) ** 2).mean(dim=red, keepdim=True)
        return (x - mean) / torch.sqrt(var + self.epsilon) * self.gamma + self.beta


class Activation(Layer):
    def __init__(self, activation, **kw):
        super().__init__(**kw)
        self.fn = _act(activation)

    def call(self, x):
        return self.fn(x)


class Concatenate(Layer):
    def __init__(self, axis=-1, **kw):
        super().__init__(**kw)
        self.axis = axis

    def call(self, xs):
        return torch.cat(list(xs), dim=self.axis)


class Flatten(Layer):
    def call(self, x):
        return x.reshape(x.shape[0], -1)


class Add(Layer):
    def call(self, xs):
        out = xs[0]
        for t in xs[1:]:
            out = out + t
        return out


def _tf_same_pad(size, k, stride):
    """TensorFlow 'SAME': out = ceil(size / stride), the odd padding element goes after"""
    out = -(-size // stride)
    total = max((out - 1) * stride + k - size, 0)
    return total // 2, total - total // 2


class Conv2D(Layer):
    """keras.layers.Conv2D, channels_last [B,H,W,C], kernel [kh,kw,C,filters]; via torch's conv2d (NCHW)"""

    def __init__(self, filters, kernel_size, strides=(1, 1), padding='valid', activation=None, use_bias=True,
                 kernel_initializer=None, **kw):
        super().__init__(**kw)
        assert tuple(strides) == (1, 1)
        self.filters, self.kernel_size, self.padding, self.activation, self.use_bias = \
            filters, tuple(kernel_size), padding, activation, use_bias

    def build(self, input_shape):
        self.kernel = self.add_weight('kernel', self.kernel_size + (input_shape[-1], self.filters))
        self.bias = self.add_weight('bias', (self.filters,)) if self.use_bias else None

    def call(self, x):
        xt = x.permute(0, 3, 1, 2)
        if self.padding == 'same':
            (t, b), (l, r) = _tf_same_pad(xt.shape[2], self.kernel_size[0], 1), _tf_same_pad(xt.shape[3], self.kernel_size[1], 1)
            xt = torch.nn.functional.pad(xt, (l, r, t, b))
        y = torch.nn.functional.conv2d(xt, self.kernel.permute(3, 2, 0, 1), self.bias)
        return _act(self.activation)(y.permute(0, 2, 3, 1))


class MaxPooling2D(Layer):
    """keras.layers.MaxPooling2D, channels_last, strides = pool_size; 'same' padding never wins the max"""

    def __init__(self, pool_size=(2, 2), strides=None, padding='valid', **kw):
        super().__init__(**kw)
        self.pool_size, self.padding = tuple(pool_size), padding
        assert strides is None

    def call(self, x):
        xt = x.permute(0, 3, 1, 2)
        if self.padding == 'same':
            (t, b), (l, r) = (_tf_same_pad(xt.shape[2], self.pool_size[0], self.pool_size[0]),
                              _tf_same_pad(xt.shape[3], self.pool_size[1], self.pool_size[1]))
            xt = torch.nn.functional.pad(xt, (l, r, t, b), value=float('-inf'))
        return torch.nn.functional.max_pool2d(xt, self.pool_size, self.pool_size).permute(0, 2, 3, 1)


class Embedding(Layer):
    def __init__(self, input_dim, output_dim, **kw):
        super().__init__(name=kw.get('name'))
        self.input_dim, self.output_dim = input_dim, output_dim

    def build(self, input_shape):
        self.embeddings = self.add_weight('embeddings', (self.input_dim, self.output_dim))

    def call(self, ids):
        return self.embeddings[_t(ids).long()]


_BOUND = {}             # keras.Input(name=...) -> the concrete seeded tensor bound to that name: the graph runs eagerly


def Input(shape=None, name=None, dtype=None, **kw):
    t = _BOUND[name]
    assert tuple(t.shape[1:]) == tuple(shape), (name, t.shape, shape)
    return t


class Model:
    def __init__(self, inputs=None, outputs=None, **kw):
        self.inputs, self.outputs = inputs, outputs

    def compile(self, *a, **kw):
        pass


class Loss:
    def __init__(self, reduction=None, name=None):
        self.reduction, self.name = reduction, name

    def __call__(self, y_true, y_pred):
        return self.call(y_true, y_pred)

    def get_config(self):
        return {}


class _Unused(Layer):
    def __init__(self, *a, **kw):
        super().__init__()


def _stub_module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    return m


class _Any:
    """attribute sink for names the layer code only touches in paths the fixtures do not take"""

    def __init__(self, *a, **kw):
        pass

    def __getattr__(self, k):
        return _Any()

    def __call__(self, *a, **kw):
        return _Any()


def install_shim():
    _install_tensor_accessors()
    tf = _make_tf()
    layers = _stub_module('keras.api.layers', Layer=Layer, Dense=Dense, Dropout=Dropout, BatchNormalization=BatchNormalization,
                          Activation=Activation, Concatenate=Concatenate, Flatten=Flatten, Input=Input, Embedding=Embedding,
                          Lambda=_Unused, Add=Add, Conv2D=Conv2D, MaxPooling2D=MaxPooling2D, SpatialDropout1D=SpatialDropout1D)
    kops = _stub_module('keras.ops', ndim=lambda x: (_t(x)).dim(), sum=tf_reduce_sum,
                        cast=lambda x, dtype: _t(x).to({'int32': torch.int32, 'int64': torch.int64,
                                                       'float32': DT}.get(dtype, dtype)),
                        log=lambda x: torch.log(_t(x)), clip=lambda x, a, b: torch.clamp(_t(x), a, b),
                        power=lambda x, p: torch.pow(_t(x), p), not_equal=lambda a, b: _t(a) != b,
                        mean=tf_reduce_mean, expand_dims=lambda x, axis: _t(x).unsqueeze(axis), split=tf_split)
    backend = _stub_module('keras.backend', epsilon=lambda: 1e-7, floatx=lambda: 'float32')

    class _Getter(types.ModuleType):
        def get(self, x):
            return x

        def serialize(self, x):
            return x
    losses = _stub_module('keras.losses', Loss=Loss, BinaryCrossentropy=_Any, MeanSquaredError=_Any,
                          CategoricalCrossentropy=_Any)
    kmodels = _stub_module('keras.api.models', Model=Model, load_model=_Any(), save_model=_Any())
    keras = _stub_module('keras', ops=kops, backend=backend, layers=layers, initializers=_Getter('keras.initializers'),
                         regularizers=_Getter('keras.regularizers'), constraints=_Getter('keras.constraints'), losses=losses,
                         optimizers=_Any())
    tf.keras = _stub_module('tensorflow.keras', layers=layers)
    ctx = _stub_module('tensorflow.python.eager.context', executing_eagerly=lambda: False, context=lambda: _Any())
    emb_ops = _stub_module('tensorflow.python.ops.embedding_ops',
                           embedding_lookup=lambda params, ids: params[_t(ids).long()])
    mods = {
        'tensorflow': tf, 'tensorflow.nn': tf.nn, 'tensorflow.keras': tf.keras,
        'tensorflow.python': _stub_module('tensorflow.python'),
        'tensorflow.python.eager': _stub_module('tensorflow.python.eager', context=ctx),
        'tensorflow.python.eager.context': ctx,
        'tensorflow.python.framework': _stub_module('tensorflow.python.framework', ops=_Any()),
        'tensorflow.python.framework.ops': _stub_module('tensorflow.python.framework.ops'),
        'tensorflow.python.keras': _stub_module('tensorflow.python.keras'),
        'tensorflow.python.keras.utils': _stub_module('tensorflow.python.keras.utils', tf_utils=_Any()),
        'tensorflow.python.ops': _stub_module('tensorflow.python.ops', embedding_ops=emb_ops, math_ops=_Any()),
        'tensorflow.python.ops.embedding_ops': emb_ops,
        'keras': keras, 'keras.ops': kops, 'keras.backend': backend,
        'keras.api': _stub_module('keras.api', layers=layers, models=kmodels), 'keras.api.models': kmodels,
        'keras.api.layers': layers, 'keras.api.metrics': _stub_module('keras.api.metrics', RootMeanSquaredError=_Any),
        'keras.initializers': keras.initializers, 'keras.regularizers': keras.regularizers,
        'keras.constraints': keras.constraints, 'keras.losses': losses,
        'keras.src': _stub_module('keras.src'), 'keras.src.legacy': _stub_module('keras.src.legacy'),
        'keras.src.legacy.losses': _stub_module('keras.src.legacy.losses', Reduction=_Any()),
    }
    sys.modules.update(mods)


def _load_file(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference_layers():
    """the reference's layers.py, unmodified, as module `deeptables.models.layers` (its package __init__ files pull in
    hypernets / pandas pipelines and are not executed; utils/consts.py and utils/counter.py are the reference's files,
    consts on a stub of the five TASK_* names it star-imports from hypernets.utils.const)"""
    logger = types.SimpleNamespace(info=lambda *a, **k: None, warning=lambda *a, **k: None, warn=lambda *a, **k: None,
                                   debug=lambda *a, **k: None, error=lambda *a, **k: None, is_info_enabled=lambda: False)
    hconst = _stub_module('hypernets.utils.const', TASK_AUTO='auto', TASK_BINARY='binary', TASK_MULTICLASS='multiclass',
                          TASK_REGRESSION='regression', TASK_MULTILABEL='multilabel')
    sys.modules.update({'hypernets': _stub_module('hypernets'), 'hypernets.utils': _stub_module('hypernets.utils', const=hconst),
                        'hypernets.utils.const': hconst,
                        'hypernets.tabular': _stub_module('hypernets.tabular', get_tool_box=_Any())})
    utils = _stub_module('deeptables.utils', dt_logging=types.SimpleNamespace(get_logger=lambda n: logger),
                         gpu=_Any(), to_dataset=_Any())
    utils.__path__ = []
    pkg = _stub_module('deeptables')
    pkg.__path__ = []
    models = _stub_module('deeptables.models')
    models.__path__ = []
    sys.modules.update({'deeptables': pkg, 'deeptables.utils': utils, 'deeptables.models': models})
    utils.consts = _load_file('deeptables.utils.consts', 'deeptables/utils/consts.py')
    utils.counter = _load_file('deeptables.utils.counter', 'deeptables/utils/counter.py')
    return _load_file('deeptables.models.layers', 'deeptables/models/layers.py')


def load_reference_deepnets():
    """the reference's deepnets.py (net functions: which layers on which of the four input tensors), unmodified"""
    sys.modules['tensorflow.python.keras.utils.generic_utils'] = _stub_module(
        'tensorflow.python.keras.utils.generic_utils', serialize_keras_object=_Any(),
        deserialize_keras_object=lambda name, module_objects=None, custom_objects=None, printable_module_name=None:
        module_objects[name])
    sys.modules['deeptables.models'].layers = sys.modules['deeptables.models.layers']
    mod = _load_file('deeptables.models.deepnets', 'deeptables/models/deepnets.py')
    sys.modules['deeptables.models'].deepnets = mod
    return mod


def load_reference_deepmodel():
    """the reference's metainfo.py, config.py and deepmodel.py, unmodified: ModelConfig (its defaults), the column
    descriptors and DeepModel, whose private __build_model (deepmodel.py:259-317) wires inputs -> embeddings -> concat ->
    BatchNormalization -> net functions -> stacking -> task_output.  keras.Input returns the tensor bound to its name
    (_BOUND), so building the graph evaluates it."""
    metainfo = _load_file('deeptables.models.metainfo', 'deeptables/models/metainfo.py')
    config = _load_file('deeptables.models.config', 'deeptables/models/config.py')
    deepmodel = _load_file('deeptables.models.deepmodel', 'deeptables/models/deepmodel.py')
    return metainfo, config, deepmodel


# ---------------------------------------------------------------------------------------------------------------
# 2. cases: (reference layer code run on the shim) vs (the oracle's restatement), same inputs and weights.
#    A case = the oracle function's name + its tensor arguments + its static arguments; the fixture stores exactly
#    that plus the reference code's output, so tests/test_oracle_reference_code.py can replay it without /root/reference.
# ---------------------------------------------------------------------------------------------------------------
def pack(prefix, v, out):
    """nested dicts / lists / tuples of tensors -> flat {key: ndarray}; None -> marker"""
    if v is None:
        out[prefix + '@none'] = np.zeros(0)
    elif isinstance(v, dict):
        out[prefix + '@keys'] = np.array(json.dumps(list(v)))
        for k, e in v.items():
            pack(f'{prefix}/{k}', e, out)
    elif isinstance(v, (list, tuple)):
        out[prefix + '@len'] = np.array(len(v))
        for k, e in enumerate(v):
            pack(f'{prefix}#{k}', e, out)
    else:
        out[prefix] = v.detach().cpu().numpy()


def layer_contract(L, outcome, zeros):
    """bad-input behaviour of the layer classes of module L (the reference's layers.py here; tests replay the same calls on
    deeptables_amd.models.layers) -> {case: {'raises': type name} | {'ok': ...}}"""
    x2, x3 = zeros(4, 6), zeros(4, 3, 2)
    mha = {'num_heads': 1, 'dropout_rate': 0, 'use_residual': True}
    cin = {'cross_layer_size': (4, 4), 'activation': 'relu', 'use_residual': False, 'use_bias': False, 'direct': False, 'reduce_D': False}
    return {
        'FM on a 2-D tensor': outcome(lambda: L.FM()(x2)),
        'MultiheadAttention on a 2-D tensor': outcome(lambda: L.MultiheadAttention(params=mha)(x2)),
        'SENET on a 2-D tensor': outcome(lambda: L.SENET(pooling_op='mean', reduction_ratio=2)(x2)),
        'BilinearInteraction on a 2-D tensor': outcome(lambda: L.BilinearInteraction(bilinear_type='field_all')(x2)),
        'Cross on a 3-D tensor': outcome(lambda: L.Cross(params={'num_cross_layer': 2})(x3)),
        'InnerProduct on 2-D embeddings': outcome(lambda: L.InnerProduct()([x2, x2])),
        'OuterProduct on 2-D embeddings': outcome(lambda: L.OuterProduct(params={'outer_product_kernel_type': 'mat'})([x2, x2])),
        'OuterProduct with an unknown kernel type': outcome(lambda: L.OuterProduct(params={'outer_product_kernel_type': 'xyz'})),
        'CIN without layers': outcome(lambda: L.CIN(params=dict(cin, cross_layer_size=()))),
        'CIN on a 2-D tensor': outcome(lambda: L.CIN(params=cin)(x2)),
        'CIN split with an odd hidden layer': outcome(lambda: L.CIN(params=dict(cin, cross_layer_size=(5, 4)))(x3)),
        'AFM on a single embedding': outcome(lambda: L.AFM(params={'dropout_rate': 0})([zeros(4, 1, 2)])),
        'AFM on 2-D embeddings': outcome(lambda: L.AFM(params={'dropout_rate': 0})([x2, x2])),
        'MultiColumnEmbedding(input_dims=5)': outcome(lambda: L.MultiColumnEmbedding(input_dims=5, output_dims=[4])),
        'MultiColumnEmbedding(output_dims=4)': outcome(lambda: L.MultiColumnEmbedding(input_dims=[5], output_dims=4)),
        'MultiColumnEmbedding with lists of different lengths': outcome(lambda: L.MultiColumnEmbedding(input_dims=[5, 6], output_dims=[4])),
        'MultiColumnEmbedding called with the wrong number of columns': outcome(
            lambda: L.MultiColumnEmbedding(input_dims=[5, 6], output_dims=[4, 4])(zeros(4, 3))),
    }


def layer_configs(L, jsonable):
    """get_config() of every layer class, constructor arguments as the net functions pass them"""
    mk = {
        'FM': lambda: L.FM(),
        'MultiheadAttention': lambda: L.MultiheadAttention(params={'num_heads': 2, 'dropout_rate': 0.1, 'use_residual': False}),
        'FGCNN': lambda: L.FGCNN(filters=3, kernel_height=4, new_filters=2, pool_height=2),
        'SENET': lambda: L.SENET(pooling_op='max', reduction_ratio=2),
        'BilinearInteraction': lambda: L.BilinearInteraction(bilinear_type='field_each'),
        'Cross': lambda: L.Cross(params={'num_cross_layer': 3}),
        'InnerProduct': lambda: L.InnerProduct(),
        'OuterProduct': lambda: L.OuterProduct(params={'outer_product_kernel_type': 'vec'}),
        'CIN': lambda: L.CIN(params={'cross_layer_size': (6, 4), 'activation': 'tanh', 'use_residual': True, 'use_bias': True,
                                     'direct': True, 'reduce_D': False}),
        'AFM': lambda: L.AFM(params={'hidden_factor': 5, 'dropout_rate': 0}),
    }
    out = {}
    for name, fn in mk.items():
        out[name] = jsonable(fn().get_config())
    return out


def main():
    global _RNG
    sys.path.insert(0, ROOT)
    install_shim()
    L = load_reference_layers()
    from oracle import reference_layers as R
    rng = np.random.RandomState(20250924)
    _RNG = rng
    written = []

    def rand(*shape):
        return torch.as_tensor(rng.randn(*shape), dtype=DT)

    def case(name, ref_out, fn, tensors, static=None, post=None):
        static = static or {}
        got = getattr(R, fn)(**tensors, **static)
        if post == 'cat1':
            got, ref_out = torch.cat(list(got), 1), torch.cat(list(ref_out), 1)
        ref_out = _t(ref_out)
        assert ref_out.shape == got.shape, (name, tuple(ref_out.shape), tuple(got.shape))
        err = (ref_out - got).abs().max().item()
        assert err < 1e-12 * max(1.0, ref_out.abs().max().item()), (name, err)
        d = {'out': ref_out.detach().cpu().numpy(),
             'meta': np.array(json.dumps({'fn': fn, 'static': static, 'post': post, 'args': sorted(tensors)}))}
        for k, v in tensors.items():
            pack('arg:' + k, v, d)
        np.savez(os.path.join(HERE, f'reference_code_{name}.npz'), **d)
        written.append(name)
        print(f'{name:30s} reference code == oracle.{fn:24s} max|diff| = {err:.1e}   out {tuple(ref_out.shape)}')

    B, F, D = 7, 5, 4
    # FM (layers.py:53-62)
    x = rand(B, F, D)
    case('fm', L.FM()(x), 'fm', {'x': x})
    # Cross (layers.py:421-436), 3 layers; zeros-initialised biases get seeded values from the shim
    xc = rand(B, 11)
    cr = L.Cross(params={'num_cross_layer': 3})
    case('cross', cr(xc), 'cross', {'x': xc, 'kernels': list(cr.kernels), 'biases': list(cr.bias)})
    # InnerProduct / OuterProduct (layers.py:473-487, :543-581): list of F x [B,1,D]
    xs = [rand(B, 1, D) for _ in range(F)]
    case('inner_product', L.InnerProduct()(xs), 'inner_product', {'xs': xs})
    for kt in ('mat', 'vec', 'num'):
        op = L.OuterProduct(params={'outer_product_kernel_type': kt})
        out = op(xs)
        case(f'outer_product_{kt}', out, 'outer_product', {'xs': xs, 'kernel': op.kernel}, {'kernel_type': kt})
    # CIN (layers.py:638-734), reduce_D = False (the oracle restates that branch; BASELINE configs use it)
    xcin = rand(B, F, D)
    for tag, params in (('split_bias', dict(cross_layer_size=(6, 4, 3), activation='relu', use_residual=False, use_bias=True,
                                            direct=False, reduce_D=False)),
                        ('direct_residual', dict(cross_layer_size=(4, 5), activation='sigmoid', use_residual=True,
                                                 use_bias=False, direct=True, reduce_D=False)),
                        ('split_linear', dict(cross_layer_size=(4, 2), activation='linear', use_residual=False, use_bias=False,
                                              direct=False, reduce_D=False))):
        cin = L.CIN(params=params)
        out = cin(xcin)
        tensors = {'x': xcin, 'filters': list(cin.f_), 'biases': list(cin.bias) if params['use_bias'] else None,
                   'dense_out': (cin.exFM_out.kernel, cin.exFM_out.bias),
                   'dense_out0': (cin.exFM_out0.kernel, cin.exFM_out0.bias) if params['use_residual'] else None}
        case(f'cin_{tag}', out, 'cin', tensors, {'cross_layer_size': list(params['cross_layer_size']),
                                                 'activation': params['activation'], 'direct': params['direct']})
    # MultiheadAttention (layers.py:104-153), training-mode BatchNormalization, dropout_rate 0
    xa = rand(B, F, 8)
    for heads, res in ((2, True), (4, False)):
        mha = L.MultiheadAttention(params={'num_heads': heads, 'dropout_rate': 0, 'use_residual': res})
        out = mha(xa)
        w = {'Q': (mha.dense_Q.kernel, mha.dense_Q.bias), 'K': (mha.dense_K.kernel, mha.dense_K.bias),
             'V': (mha.dense_V.kernel, mha.dense_V.bias), 'bn': (mha.batch_normalize.gamma, mha.batch_normalize.beta)}
        if res:
            w['R'] = (mha.dense_residual.kernel, mha.dense_residual.bias)
        # the oracle takes the weights as ONE dict argument: stored as separate tensors, rebuilt by the replay
        names = sorted(w)
        case(f'mha_h{heads}_res{int(res)}', out, '_mha_from_parts',
             {'x': xa, 'parts': [list(w[n]) for n in names]}, {'names': names, 'num_heads': heads, 'use_residual': res})
    # MultiColumnEmbedding (layers.py:853-904): float ids are cast (truncated) to int32, one table per column
    dims = [5, 9, 3]
    mce = L.MultiColumnEmbedding(input_dims=dims, output_dims=[D] * 3)
    ids = torch.as_tensor(np.stack([rng.randint(0, d, size=B) for d in dims], 1).astype(np.float32) + 0.4, dtype=torch.float32)
    outs = mce(ids)
    case('multi_column_embedding', outs, 'multi_column_embedding', {'inputs': ids, 'tables': list(mce.embeddings)}, post='cat1')
    # AFM (layers.py:742-812)
    afm = L.AFM(params={'hidden_factor': 3, 'dropout_rate': 0})
    out = afm(xs)
    case('afm', out, 'afm', {'xs': xs, 'att_kernel': afm.dense_attention.kernel, 'att_bias': afm.dense_attention.bias,
                             'projection_h': afm.attention_p, 'out_kernel': afm.dense_out.kernel}, {'activation': 'relu'})
    # BilinearInteraction (layers.py:311-382) and SENET (:245-308) take the stacked [B,F,D] block
    x3 = rand(B, F, D)
    for bt in ('field_all', 'field_each', 'field_interaction'):
        bi = L.BilinearInteraction(bilinear_type=bt)
        out = bi(x3)
        case(f'bilinear_{bt}', out, 'bilinear_interaction', {'x': x3, 'W_list': [bi.W] if bt == 'field_all' else list(bi.W_list)},
             {'bilinear_type': bt})
    for pool in ('mean', 'max'):
        se = L.SENET(pooling_op=pool, reduction_ratio=2)
        out = se(x3)
        case(f'senet_{pool}', out, 'senet', {'x': x3, 'att1': (se.dense_att1.kernel, se.dense_att1.bias),
                                             'att2': (se.dense_att2.kernel, se.dense_att2.bias)}, {'pooling_op': pool})
    # ---- the net functions of deepnets.py (which layers read which of the four graph tensors, in which order) against
    #      the oracle's model_nets — the part of DeepModel.__build_model that feeds them (embedding lookup, Flatten /
    #      Concatenate, BatchNormalization of concat[flatten_emb, dense], deepmodel.py:259-274, 348-361) is formed here
    #      with the same shim layers, since that method cannot be imported without the reference's whole package
    N = load_reference_deepnets()
    Bn, Fn, Dn, Nd = 9, 4, 6, 3
    dims = [7, 5, 11, 4]
    mce = L.MultiColumnEmbedding(input_dims=dims, output_dims=[Dn] * Fn)
    idx = torch.as_tensor(np.stack([rng.randint(0, d, size=Bn) for d in dims], 1).astype(np.float32), dtype=torch.float32)
    embeddings = mce(idx)                                                        # list of F x [B,1,D]
    dense = rand(Bn, Nd)
    flatten_emb = Flatten()(Concatenate(axis=-1)(embeddings))                    # deepmodel.py:269-274
    bn = BatchNormalization(name='bn_concat_emb_dense')
    concat_emb_dense = bn(Concatenate()([flatten_emb, dense]))                   # deepmodel.py:348-361
    config = types.SimpleNamespace(
        dnn_params={'hidden_units': ((16, 0, False), (8, 0, False)), 'activation': 'relu'},
        cross_params={'num_cross_layer': 3},
        cin_params={'cross_layer_size': (6, 4), 'activation': 'relu', 'use_residual': False, 'use_bias': False,
                    'direct': False, 'reduce_D': False},
        autoint_params={'num_attention': 2, 'num_heads': 2, 'dropout_rate': 0, 'use_residual': True})
    desc = types.SimpleNamespace(add_net=lambda *a, **k: None)
    ocfg = {'cin_params': dict(config.cin_params, cross_layer_size=list(config.cin_params['cross_layer_size'])),
            'autoint_params': config.autoint_params, 'dnn_activation': 'relu'}

    def by_name(layers, name):
        return next(l for l in layers if l.name == name)

    def run_net(net):
        start = len(_REGISTRY)
        out = getattr(N, net)(embeddings, flatten_emb, dense, concat_emb_dense, config, desc)
        new = _REGISTRY[start:]
        names, parts = [], []
        if net == 'linear':
            names, parts = ['linear_logit'], [by_name(new, 'linear_logit').kernel]
        elif net in ('dnn_nets', 'dcn_nets'):
            pre = 'dnn' if net == 'dnn_nets' else 'dcn'
            ds = [by_name(new, f'{pre}_dense_{i}') for i in (1, 2)]
            names, parts = ['dnn' if net == 'dnn_nets' else 'dcn_dnn'], [[[d.kernel, d.bias] for d in ds]]
            if net == 'dcn_nets':
                cr_ = by_name(new, 'dcn_cross_layer')
                names += ['dcn_cross_kernels', 'dcn_cross_bias']
                parts += [list(cr_.kernels), list(cr_.bias)]
        elif net == 'cin_nets':
            c = next(l for l in new if isinstance(l, L.CIN))
            names, parts = ['cin_filters', 'cin_exFM_out'], [list(c.f_), [c.exFM_out.kernel, c.exFM_out.bias]]
        elif net == 'autoint_nets':
            ms = [l for l in new if isinstance(l, L.MultiheadAttention)]
            names = ['autoint_layers']
            parts = [[[[m.dense_Q.kernel, m.dense_Q.bias], [m.dense_K.kernel, m.dense_K.bias],
                       [m.dense_V.kernel, m.dense_V.bias], [m.dense_residual.kernel, m.dense_residual.bias],
                       [m.batch_normalize.gamma, m.batch_normalize.beta]] for m in ms]]
        case(f'net_{net}', out, '_nets_from_parts',
             {'cat_idx': idx, 'dense': dense, 'tables': list(mce.embeddings), 'bn': [bn.gamma, bn.beta], 'parts': parts},
             {'names': names, 'nets': [net], 'config': ocfg, 'net': net})

    for net in ('linear', 'fm_nets', 'dnn_nets', 'dcn_nets', 'cin_nets', 'autoint_nets'):
        run_net(net)
    # ---- the remaining layer types of layers.py: FGCNN (:161-242), VarLenColumnEmbedding (:925-980), the losses (:983-1163)
    x4 = rand(B, 6, D, 1)
    fg = L.FGCNN(filters=3, kernel_height=4, new_filters=2, pool_height=2)
    pooled, feats = fg(x4)
    case('fgcnn', torch.cat([pooled.reshape(B, -1), feats.reshape(B, -1)], -1), '_fgcnn_from_parts',
         {'x': x4, 'conv_kernel': fg.conv2d.kernel, 'conv_bias': fg.conv2d.bias, 'dense_kernel': fg.dense_output.kernel,
          'dense_bias': fg.dense_output.bias}, {'pool_height': 2, 'new_filters': 2})
    fg2 = L.FGCNN(filters=2, kernel_height=7, new_filters=1, pool_height=3, activation='relu')   # taller than the 5 fields
    x5 = rand(B, 5, D, 2)
    pooled, feats = fg2(x5)
    case('fgcnn_tall_kernel', torch.cat([pooled.reshape(B, -1), feats.reshape(B, -1)], -1), '_fgcnn_from_parts',
         {'x': x5, 'conv_kernel': fg2.conv2d.kernel, 'conv_bias': fg2.conv2d.bias, 'dense_kernel': fg2.dense_output.kernel,
          'dense_bias': fg2.dense_output.bias}, {'pool_height': 3, 'new_filters': 1, 'activation': 'relu'})
    vl = L.VarLenColumnEmbedding(emb_vocab_size=11, emb_output_dim=3, embeddings_initializer='uniform',
                                 embeddings_regularizer=None, activity_regularizer=None, dropout_rate=0.)
    vids = torch.as_tensor(rng.randint(0, 11, size=(B, 4)).astype(np.float32))
    case('var_len_column_embedding', vl(vids), 'var_len_embedding', {'inputs': vids, 'table': vl.emb_layer.embeddings})
    y01 = torch.as_tensor((rng.rand(B, 1) < 0.4).astype(np.float64))
    pr = torch.as_tensor(rng.uniform(0.02, 0.98, size=(B, 1)))
    pr[0, 0], pr[1, 0] = 0.0, 1.0                                                  # the clip to [eps, 1 - eps]
    case('binary_focal_loss', L.BinaryFocalLoss(gamma=1.5, alpha=0.3).call(y01, pr.clone()), 'binary_focal_loss',
         {'y_true': y01, 'y_pred': pr}, {'gamma': 1.5, 'alpha': 0.3})
    yc = torch.as_tensor(np.eye(3)[rng.randint(0, 3, size=B)])
    pc = torch.as_tensor(rng.uniform(0.05, 1.0, size=(B, 3)))                      # not normalised: the loss rescales
    case('categorical_focal_loss', L.CategoricalFocalLoss().call(yc, pc.clone()), 'categorical_focal_loss',
         {'y_true': yc, 'y_pred': pc})
    gh = L.GHMCLoss(bins=6, momentum=0.6)
    zin, tgt = rand(B, 2) * 2, torch.as_tensor((rng.rand(B, 2) < 0.5).astype(np.float64))
    acc0 = gh.acc_sum.clone()
    loss1 = gh.calc(zin, tgt)
    case('ghmc_loss_step1', torch.cat([loss1.reshape(1), gh.acc_sum.reshape(-1)]), '_ghmc_from_parts',
         {'input': zin, 'target': tgt, 'acc_sum': acc0}, {'bins': 6, 'momentum': 0.6})
    acc1, zin2 = gh.acc_sum.clone(), rand(B, 2)
    loss2 = gh.calc(zin2, tgt)                                                    # second call: the moving bin counts
    case('ghmc_loss_step2', torch.cat([loss2.reshape(1), gh.acc_sum.reshape(-1)]), '_ghmc_from_parts',
         {'input': zin2, 'target': tgt, 'acc_sum': acc1}, {'bins': 6, 'momentum': 0.6})
    gh0 = L.GHMCLoss(bins=5, momentum=0)
    loss0 = gh0.calc(zin, tgt)
    case('ghmc_loss_no_momentum', torch.cat([loss0.reshape(1), torch.zeros(5, dtype=DT)]), '_ghmc_from_parts',
         {'input': zin, 'target': tgt, 'acc_sum': torch.zeros(5, dtype=DT)}, {'bins': 5, 'momentum': 0})

    # ---- whole models: the reference's DeepModel.__build_model (deepmodel.py:259-317), ModelConfig defaults (config.py) and
    #      column descriptors (metainfo.py), all unmodified.  `nets` is handed to __build_model in list order (ModelConfig
    #      passes it through set(), whose order changes from process to process; the order only matters for 'concat').
    M, C, DM = load_reference_deepmodel()

    def reference_weights(layers):
        by = {l.name: l for l in layers if getattr(l, 'name', None)}
        w = {}
        mce_ = next((l for l in layers if isinstance(l, L.MultiColumnEmbedding)), None)
        w['emb_categorical_vars_all'] = list(mce_.embeddings) if mce_ is not None else []
        vls = [l for l in layers if isinstance(l, L.VarLenColumnEmbedding)]
        if vls:
            w['var_len_tables'] = [l.emb_layer.embeddings for l in vls]
        if 'bn_concat_emb_dense' in by:
            w['bn_concat_emb_dense'] = [by['bn_concat_emb_dense'].gamma, by['bn_concat_emb_dense'].beta]
        if 'linear_logit' in by:
            w['linear_logit'] = by['linear_logit'].kernel

        def cells(prefix):
            out, i = [], 1
            while f'{prefix}_dense_{i}' in by:
                d_, bn_ = by[f'{prefix}_dense_{i}'], by.get(f'{prefix}_bn_{i}')
                out.append([d_.kernel, d_.bias] if bn_ is None else [d_.kernel, None, [bn_.gamma, bn_.beta]])
                i += 1
            return out
        for prefix, key in (('dnn', 'dnn'), ('dcn', 'dcn_dnn'), ('opnn', 'opnn'), ('ipnn', 'ipnn'), ('pnn', 'pnn'),
                            ('cross_dnn', 'cross_dnn'), ('fibi_dnn', 'fibi_dnn'), ('fgcnn_dnn', 'fgcnn_dnn'),
                            ('fgcnn_ipnn', 'fgcnn_ipnn')):
            if f'{prefix}_dense_1' in by:
                w[key] = cells(prefix)
        for l in layers:
            n = getattr(l, 'name', None) or ''
            if n.startswith('dense_logit_'):
                w[n] = l.kernel
            if isinstance(l, L.CIN):
                w['cin_filters'], w['cin_exFM_out'] = list(l.f_), [l.exFM_out.kernel, l.exFM_out.bias]
                if l.use_bias:
                    w['cin_bias'] = list(l.bias)
            elif isinstance(l, L.Cross):
                key = {'dcn_cross_layer': 'dcn_cross', 'cross_layer': 'cross', 'cross_dnn_layer': 'cross_dnn'}[n]
                w[key + '_kernels'], w[key + '_bias'] = list(l.kernels), list(l.bias)
            elif isinstance(l, L.OuterProduct):
                w[{'outer_product_layer': 'opnn_kernel', 'pnn_outer_product_layer': 'pnn_kernel'}[n]] = l.kernel
            elif isinstance(l, L.AFM):
                w.setdefault('afm', []).append({'att_kernel': l.dense_attention.kernel, 'att_bias': l.dense_attention.bias,
                                                'projection_h': l.attention_p, 'out_kernel': l.dense_out.kernel})
            elif isinstance(l, L.SENET):
                w.setdefault('senet', []).append({'att1': [l.dense_att1.kernel, l.dense_att1.bias],
                                                  'att2': [l.dense_att2.kernel, l.dense_att2.bias]})
            elif isinstance(l, L.BilinearInteraction):
                Ws = [l.W] if l.bilinear_type == 'field_all' else list(l.W_list)
                w.setdefault('bilinear', {})['senet' if n.startswith('senet_bilinear') else 'embedding'] = Ws
            elif isinstance(l, L.FGCNN):
                w.setdefault('fgcnn', []).append({'conv_kernel': l.conv2d.kernel, 'conv_bias': l.conv2d.bias,
                                                  'dense_kernel': l.dense_output.kernel, 'dense_bias': l.dense_output.bias})
            elif isinstance(l, L.MultiheadAttention):
                lw = {'Q': [l.dense_Q.kernel, l.dense_Q.bias], 'K': [l.dense_K.kernel, l.dense_K.bias],
                      'V': [l.dense_V.kernel, l.dense_V.bias], 'bn': [l.batch_normalize.gamma, l.batch_normalize.beta]}
                if l.use_residual:
                    lw['R'] = [l.dense_residual.kernel, l.dense_residual.bias]
                w.setdefault('autoint_layers', []).append(lw)
        to = by['task_output']
        w['task_output'] = [to.kernel, to.bias]
        return w

    graph_names = {}

    def run_model(tag, nets, task='binary', num_classes=2, vocab=(7, 5, 11, 4, 6), emb_dim=4, n_dense=3, batch=8,
                  init_scale=1.0, var_len=(), **conf):
        global _INIT_SCALE
        _INIT_SCALE = init_scale
        conf.setdefault('embedding_dropout', 0)                 # config.py:84 default 0.3: dropout is outside a value pin
        config = C.ModelConfig(nets=nets, embeddings_output_dim=emb_dim, **conf)
        assert sorted(config.nets) == sorted(nets)
        cats = [M.CategoricalColumn(f'C{i}', v, emb_dim) for i, v in enumerate(vocab)]
        conts = [M.ContinuousColumn('input_continuous_all', [f'I{j}' for j in range(n_dense)], input_dim=n_dense)] \
            if n_dense else []
        ids = torch.as_tensor(np.stack([rng.randint(0, v, size=batch) for v in vocab], 1).astype(np.float32))
        dn = rand(batch, n_dense) if n_dense else None
        _BOUND.clear()
        _BOUND['input_categorical_vars_all'] = ids
        if n_dense:
            _BOUND['input_continuous_all'] = dn
        vl_cols, vl_ids = [], []
        for k, (vl_vocab, vl_len) in enumerate(var_len):            # variable-length columns (deepmodel.py:375-378, 406-418)
            col = M.VarLenCategoricalColumn(f'tags{k}', vl_vocab, emb_dim)
            col.max_elements_length = vl_len                        # set by the preprocessor in the reference (metainfo.py:69)
            vl_cols.append(col)
            vl_ids.append(torch.as_tensor(rng.randint(0, vl_vocab, size=(batch, vl_len)).astype(np.float32)))
            _BOUND[col.name] = vl_ids[-1]
        start = len(_REGISTRY)
        dm = DM.DeepModel(task, num_classes, config, cats, conts, var_categorical_len_columns=vl_cols or None)
        model = dm._DeepModel__build_model(task=task, num_classes=num_classes, nets=list(nets), categorical_columns=cats,
                                           continuous_columns=conts, var_len_categorical_columns=vl_cols or None,
                                           config=config)
        layers_ = _REGISTRY[start:]
        graph_names[tag] = [l.name for l in layers_ if l.name]          # the layers the reference names explicitly, in order
        head = next(l for l in layers_ if l.name == 'task_output')
        ocfg = {'cin_params': dict(config.cin_params, cross_layer_size=list(config.cin_params['cross_layer_size'])),
                'autoint_params': config.autoint_params, 'fibinet_params': config.fibinet_params,
                'fgcnn_params': {k: list(v) for k, v in config.fgcnn_params.items()}, 'pnn_params': config.pnn_params,
                'dnn_activation': config.dnn_params.get('activation', 'relu'), 'stacking_op': config.stacking_op,
                'task': task,
                # not read by the oracle: what a caller needs to build the same model through the ModelConfig API
                'build': {'num_classes': num_classes, 'embeddings_output_dim': emb_dim, 'output_use_bias': config.output_use_bias,
                          'dnn_params': dict(config.dnn_params, hidden_units=[list(h) for h in config.dnn_params['hidden_units']]),
                          'cross_params': config.cross_params, 'afm_params': config.afm_params}}
        wnest = reference_weights(layers_)
        extra = {'var_len_idx': vl_ids} if vl_ids else {}
        if vl_ids:
            ocfg['build']['var_len'] = [[c.name, c.vocabulary_size, c.max_elements_length] for c in vl_cols]
        case(f'model_{tag}', torch.cat([head.last_preact, model.outputs], -1), '_model_from_parts',
             {'cat_idx': ids, 'dense': dn, 'weights': wnest, **extra}, {'nets': list(nets), 'config': ocfg})
        # the gradients of the task's Keras loss (deepmodel.py:319-346; the documented formulas on the model OUTPUT, with
        # Keras' clip of the probabilities to [1e-7, 1 - 1e-7]) through the reference's graph, by autograd on the shim's ops
        out = model.outputs
        if task == 'binary':
            yv = torch.as_tensor((rng.rand(batch, 1) < 0.3).astype(np.float64))
            pc = torch.clamp(out, 1e-7, 1 - 1e-7)
            loss = -(yv * torch.log(pc) + (1 - yv) * torch.log(1 - pc)).mean()
        elif task == 'regression':
            yv = rand(batch, 1) * 2
            loss = ((out - yv) ** 2).mean()
        else:
            yv = torch.as_tensor(np.eye(num_classes)[rng.randint(0, num_classes, size=batch)])
            pc = torch.clamp(out / out.sum(-1, keepdim=True), 1e-7, 1 - 1e-7)
            loss = -(yv * torch.log(pc)).sum(-1).mean()
        leaves = R._leaves(wnest)
        grads = torch.autograd.grad(loss, leaves, allow_unused=True)
        flat = torch.cat([(torch.zeros_like(t) if g is None else g).reshape(-1) for t, g in zip(leaves, grads)])
        assert task != 'binary' or head.last_preact.abs().max().item() < 12, 'saturated sigmoid: Keras clips p at 1e-7'
        case(f'modelgrad_{tag}', flat, '_model_grads_from_parts',
             {'cat_idx': ids, 'dense': dn, 'weights': wnest, 'y': yv, **extra}, {'nets': list(nets), 'config': ocfg})
        _INIT_SCALE = 1.0

    small = {'hidden_units': ((12, 0, False), (6, 0, False)), 'activation': 'relu'}
    # the five BASELINE.json configurations, at fixture size
    run_model('fm', ['linear', 'fm_nets'])                                                         # configs[0]
    run_model('deepfm', N.DeepFM)                                                                  # configs[1], default 128-64 tower
    run_model('xdeepfm', N.xDeepFM, dnn_params=small,
              cin_params={'cross_layer_size': (8, 8, 8), 'activation': 'relu', 'use_residual': False, 'use_bias': False,
                          'direct': False, 'reduce_D': False})                                     # configs[2]
    run_model('autoint', N.AutoInt, emb_dim=8,
              autoint_params={'num_attention': 3, 'num_heads': 4, 'dropout_rate': 0, 'use_residual': True})   # configs[3]
    run_model('dcn', N.DCN, dnn_params=small, cross_params={'num_cross_layer': 6}, init_scale=0.4)                 # configs[4]
    # the other presets of deepnets.py:14-23 and every remaining net function
    run_model('widedeep', N.WideDeep, dnn_params=small)
    run_model('pnn', N.PNN, dnn_params=small)
    run_model('afm', N.AFM)
    run_model('fibinet', N.FiBiNet, dnn_params=small)
    run_model('fgcnn', N.FGCNN, dnn_params=small)
    run_model('opnn_ipnn_vec', ['opnn_nets', 'ipnn_nets'], dnn_params=small, pnn_params={'outer_product_kernel_type': 'vec'})
    run_model('cross_and_cross_dnn', ['cross_nets', 'cross_dnn_nets'], dnn_params=small, cross_params={'num_cross_layer': 2})
    run_model('fibi_nets_flattened', ['fibi_nets', 'linear'],
              fibinet_params={'senet_pooling_op': 'max', 'senet_reduction_ratio': 2, 'bilinear_type': 'field_each'})
    run_model('fibi_nets_alone', ['fibi_nets'],
              fibinet_params={'senet_pooling_op': 'mean', 'senet_reduction_ratio': 3, 'bilinear_type': 'field_all'})
    run_model('fgcnn_cin_fm', ['fgcnn_cin_nets', 'fgcnn_fm_nets'],                     # two fg_nets calls: two FGCNN stacks
              cin_params={'cross_layer_size': (6, 4), 'activation': 'relu', 'use_residual': False, 'use_bias': False,
                          'direct': False, 'reduce_D': False},
              fgcnn_params={'fg_filters': (3, 2), 'fg_heights': (3, 2), 'fg_pool_heights': (2, 2), 'fg_new_feat_filters': (2, 1)})
    run_model('fgcnn_afm_ipnn', ['fgcnn_afm_nets', 'fgcnn_ipnn_nets'], dnn_params=small,
              fgcnn_params={'fg_filters': (2,), 'fg_heights': (3,), 'fg_pool_heights': (2,), 'fg_new_feat_filters': (2,)})
    # the rest of __build_model / __output_layer: stacking 'concat', no output bias, regression and multiclass heads,
    # a tower with BatchNormalization cells (Dense without bias -> BN -> activation), no continuous inputs
    run_model('deepfm_concat_nobias', N.DeepFM, dnn_params=small, stacking_op='concat', output_use_bias=False)
    run_model('deepfm_regression', N.DeepFM, task='regression', num_classes=None, dnn_params=small)
    run_model('dnn_multiclass', ['dnn_nets'], task='multiclass', num_classes=3, dnn_params=small)
    run_model('deepfm_bn_tower', N.DeepFM, dnn_params={'hidden_units': ((12, 0, True), (6, 0, True)), 'activation': 'tanh'})
    run_model('deepfm_no_dense', N.DeepFM, n_dense=0, dnn_params=small)
    # two variable-length categorical columns next to the fixed ones: their [B,1,L*D] blocks join the embedding list, so only
    # nets that read the flattened concatenation apply (deepmodel.py:406-418)
    run_model('dnn_var_len', ['dnn_nets'], dnn_params=small, var_len=((9, 3), (6, 5)))
    # (appended after every other case: the fixtures above keep the random stream they were verified on the GPU with)
    xcin2 = rand(B, F, D)
    # reduce_D = True (layers.py:652-655, 696-701): every layer's filter is the product of two low-rank factors
    for tag, params in (('reduce_D_split', dict(cross_layer_size=(6, 4), activation='relu', use_residual=False, use_bias=True,
                                               direct=False, reduce_D=True)),
                        ('reduce_D_direct', dict(cross_layer_size=(3, 5), activation='tanh', use_residual=False, use_bias=False,
                                                direct=True, reduce_D=True))):
        cin = L.CIN(params=params)
        out = cin(xcin2)
        tensors = {'x': xcin2, 'filters': None, 'biases': list(cin.bias) if params['use_bias'] else None,
                   'dense_out': (cin.exFM_out.kernel, cin.exFM_out.bias), 'dense_out0': None,
                   'reduce_factors': [[a, b_] for a, b_ in zip(cin.f0_, cin.f__)]}
        case(f'cin_{tag}', out, 'cin', tensors, {'cross_layer_size': list(params['cross_layer_size']),
                                                 'activation': params['activation'], 'direct': params['direct']})
    # (round 3, appended after everything above so earlier fixtures keep their random stream) whole models at the embedding
    # widths the product's FAST kernels take: AutoInt at D = 16 and D = 32 lands on csrc/autoint.hip (the MFMA
    # interacting-layer kernel; D = 8 above runs the VALU fallback of mha.hip), xDeepFM at D = 16 is replayed in the
    # bf16-MFMA mode of the CIN (csrc/cin_bf16.hip) at north_star's 1e-2
    run_model('autoint_d16', N.AutoInt, emb_dim=16, dnn_params=small,
              autoint_params={'num_attention': 2, 'num_heads': 2, 'dropout_rate': 0, 'use_residual': True})
    run_model('autoint_d32', N.AutoInt, emb_dim=32, dnn_params=small, vocab=(7, 5, 11, 4, 6, 9, 3),
              autoint_params={'num_attention': 3, 'num_heads': 4, 'dropout_rate': 0, 'use_residual': True})
    run_model('xdeepfm_d16', N.xDeepFM, emb_dim=16, dnn_params=small, batch=16,
              cin_params={'cross_layer_size': (16, 12, 8), 'activation': 'relu', 'use_residual': False, 'use_bias': False,
                          'direct': False, 'reduce_D': False})
    # ---- the plugin / configuration surface: what the reference's own config.py / metainfo.py / deepnets.py
    #      answer, recorded as JSON for tests/test_oracle_reference_code.py::test_drop_in_api_answers_like_the_reference
    def outcome(fn):
        try:
            return {'ok': jsonable(fn())}
        except Exception as e:                      # the exception TYPE is the contract (messages are free text)
            return {'raises': type(e).__name__}

    def jsonable(v):
        if isinstance(v, (list, tuple)):
            return [jsonable(e) for e in v]
        if isinstance(v, dict):
            return {str(k): jsonable(e) for k, e in v.items()}
        if callable(v):
            return getattr(v, '__name__', 'callable')
        return v if isinstance(v, (str, int, float, bool, type(None))) else repr(v)

    def good_net(embeddings, flatten_emb_layer, dense_layer, concat_emb_dense, config, model_desc):
        return None

    def bad_net(embeddings, dense_layer):
        return None

    cfg0 = C.ModelConfig()
    api = {
        'ModelConfig_fields': list(cfg0._fields),
        'ModelConfig_defaults': {k: jsonable(getattr(cfg0, k)) for k in cfg0._fields if k != 'nets'},
        'ModelConfig_default_nets': sorted(cfg0.nets),
        'presets': {k: list(getattr(N, k)) for k in ('WideDeep', 'DeepFM', 'xDeepFM', 'AutoInt', 'DCN', 'FGCNN', 'FiBiNet', 'PNN', 'AFM')},
        'net_functions': sorted(n for n in dir(N) if callable(getattr(N, n)) and not n.startswith('_') and
                                getattr(getattr(N, n), '__module__', '') == N.__name__ and
                                inspect.isfunction(getattr(N, n)) and
                                inspect.signature(getattr(N, n)) == inspect.signature(N.linear)),
        'CategoricalColumn': jsonable(M.CategoricalColumn('c', 10)._asdict()),
        'VarLenCategoricalColumn': jsonable(M.VarLenCategoricalColumn('v', 10)._asdict()),
        'ContinuousColumn': jsonable(M.ContinuousColumn('x', ['a', 'b'])._asdict()),
        'behaviour': {
            'get(None)': outcome(lambda: N.get(None)),
            'get(123)': outcome(lambda: N.get(123)),
            "get('dnn_nets')": outcome(lambda: N.get('dnn_nets')),
            'get(callable with the plugin signature)': outcome(lambda: N.get(good_net)),
            'get(callable with another signature)': outcome(lambda: N.get(bad_net)),
            "register_nets('not callable')": outcome(lambda: N.register_nets('x')),
            'register_nets(plugin)': outcome(lambda: N.register_nets(good_net)),
            'get_nets(names + plugin), sorted': outcome(lambda: sorted(N.get_nets(['linear', good_net, 'linear']))),
            'ModelConfig(var_len item of length 2)': outcome(lambda: C.ModelConfig(var_len_categorical_columns=[('g', '|')])),
            'ModelConfig(var_len column also excluded)': outcome(
                lambda: C.ModelConfig(exclude_columns=['g'], var_len_categorical_columns=[('g', '|', 'max')])),
            'ModelConfig(var_len column also categorical)': outcome(
                lambda: C.ModelConfig(categorical_columns=['g'], var_len_categorical_columns=[('g', '|', 'max')])),
            'ModelConfig(var_len ok).var_len_categorical_columns': outcome(
                lambda: C.ModelConfig(var_len_categorical_columns=[('g', '|', 'max')]).var_len_categorical_columns),
            'hash(ModelConfig) == hash(name)': outcome(lambda: hash(C.ModelConfig(name='abc')) == hash('abc')),
        },
    }
    def build_outcome(nets, task='binary', num_classes=2, n_cat=3, n_dense=2, **conf):
        def go():
            config = C.ModelConfig(nets=nets, embedding_dropout=0, **conf)
            cats = [M.CategoricalColumn(f'C{i}', 5 + i, 4) for i in range(n_cat)]
            conts = [M.ContinuousColumn('input_continuous_all', [f'I{j}' for j in range(n_dense)], input_dim=n_dense)] \
                if n_dense else []
            _BOUND.clear()
            _BOUND['input_categorical_vars_all'] = torch.zeros(4, n_cat, dtype=torch.float32)
            _BOUND['input_continuous_all'] = rand(4, max(n_dense, 1))[:, :n_dense]
            dm = DM.DeepModel(task, num_classes, config, cats, conts)
            out = dm._DeepModel__build_model(task=task, num_classes=num_classes, nets=list(nets), categorical_columns=cats,
                                             continuous_columns=conts, var_len_categorical_columns=None, config=config).outputs
            return list(out.shape)
        return outcome(go)
    api['build'] = {
        'DeepFM, binary': build_outcome(N.DeepFM),
        'DeepFM, multiclass 3': build_outcome(N.DeepFM, task='multiclass', num_classes=3),
        'DeepFM, multilabel 4': build_outcome(N.DeepFM, task='multilabel', num_classes=4),
        'DeepFM, regression': build_outcome(N.DeepFM, task='regression', num_classes=None),
        "stacking_op 'bogus'": build_outcome(N.DeepFM, stacking_op='bogus'),
        "task 'weird'": build_outcome(N.DeepFM, task='weird'),
        'multiclass without num_classes': build_outcome(N.DeepFM, task='multiclass', num_classes=None),
        'no inputs at all': build_outcome(['dnn_nets'], n_cat=0, n_dense=0),
        'fm_nets without categorical columns': build_outcome(['fm_nets'], n_cat=0),
        'afm_nets with one categorical column': build_outcome(['afm_nets'], n_cat=1),
        'pnn_nets with one categorical column, plus dnn_nets': build_outcome(['pnn_nets', 'dnn_nets'], n_cat=1),
        'dnn_nets on continuous inputs only': build_outcome(['dnn_nets'], n_cat=0),
        'dnn_nets on categorical inputs only': build_outcome(['dnn_nets'], n_dense=0),
        'linear on continuous inputs only': build_outcome(['linear'], n_cat=0),
    }
    # the layer classes' own contract (layers.py): dimension checks raise ValueError, get_config() carries the constructor
    # arguments (what load_model(custom_objects=dt_custom_objects) rebuilds a layer from)
    def loss_outcome(task, num_classes):
        def go():
            config = C.ModelConfig(nets=['dnn_nets'], embedding_dropout=0)
            cats = [M.CategoricalColumn('C0', 5, 4)]
            _BOUND.clear()
            _BOUND['input_categorical_vars_all'] = torch.zeros(4, 1, dtype=torch.float32)
            dm = DM.DeepModel(task, num_classes, config, cats, [])
            dm._DeepModel__build_model(task=task, num_classes=num_classes, nets=['dnn_nets'], categorical_columns=cats,
                                       continuous_columns=[], var_len_categorical_columns=None, config=config)
            return dm.model_desc.loss
        return outcome(go)
    # DeepModel.__compile_model (deepmodel.py:319-346), loss='auto': which Keras loss a task gets
    api['auto_loss'] = {f'{t}/{n}': loss_outcome(t, n) for t, n in (('binary', 2), ('multilabel', 3), ('regression', None),
                                                                  ('multiclass', 3), ('multiclass', 2))}
    api['graph_layer_names'] = graph_names        # names with a per-process counter suffix (_0, _1, ..) are compared without it
    api['layer_errors'] = layer_contract(L, outcome, lambda *shape: torch.zeros(*shape, dtype=DT))
    api['layer_get_config'] = layer_configs(L, jsonable)
    api['dt_custom_objects'] = sorted(L.dt_custom_objects)
    with open(os.path.join(HERE, 'reference_code_api.json'), 'w') as f:
        json.dump(api, f, indent=1, sort_keys=True)
    print('reference_code_api.json written')
    print(f'{len(written)} fixtures written to tests/golden/reference_code_*.npz')


if __name__ == '__main__':
    main()
