# -*- coding:utf-8 -*-
"""A small Keras-functional-API look-alike on torch, just enough to keep DeepTables' plugin
boundary intact: net functions (`deepnets.linear` signature, deeptables/models/deepnets.py:43)
receive SYMBOLIC tensors, call layers on them (`Dense(1, name='linear_logit')(x)`), and
`Model(inputs, outputs)` executes the recorded graph eagerly on torch(HIP) tensors.

Symbolic mode records (layer, inputs) nodes and infers shapes through
`Layer.compute_output_shape`; execution mode runs `layer.call(...)`.  Weights keep the Keras
names (`kernel`, `bias`, `gamma`, `beta`, `moving_mean`, `moving_variance`, `embeddings_i`) and
the Keras layouts (Dense kernel is [in, out]) so a Keras checkpoint maps 1:1.

Built-in Keras layers used by the reference graph (Dense, BatchNormalization, Concatenate, Add,
Flatten, Dropout, Activation) are library-level plumbing: Dense is one torch.addmm
(rocBLAS/hipBLASLt); BatchNormalization calls the HIP kernel with Keras semantics.
"""
import math
import re
from collections import OrderedDict

import numpy as np
import torch
from torch import nn

from . import ops

# ---------------------------------------------------------------------------------------------
# naming (keras.backend.get_uid look-alike)
# ---------------------------------------------------------------------------------------------
_UIDS = {}


def reset_uids():
    _UIDS.clear()


def _to_snake(name):
    s = re.sub('(.)([A-Z][a-z0-9]+)', r'\1_\2', name)
    return re.sub('([a-z])([A-Z])', r'\1_\2', s).lower()


def unique_name(prefix):
    n = _UIDS.get(prefix, 0)
    _UIDS[prefix] = n + 1
    return prefix if n == 0 else f'{prefix}_{n}'


# ---------------------------------------------------------------------------------------------
# initializers (Keras names)
# ---------------------------------------------------------------------------------------------
_SEED = [20241218]


def set_seed(seed):
    _SEED[0] = int(seed)


def _rng():
    _SEED[0] += 1
    return np.random.default_rng(_SEED[0])


def _fans(shape):
    if len(shape) < 1:
        return 1, 1
    if len(shape) == 1:
        return shape[0], shape[0]
    if len(shape) == 2:
        return shape[0], shape[1]
    rf = int(np.prod(shape[:-2]))
    return shape[-2] * rf, shape[-1] * rf


def initialize(shape, initializer):
    """-> float32 torch tensor initialised like the Keras initializer of that name."""
    shape = tuple(int(s) for s in shape)
    if callable(initializer):
        return torch.as_tensor(np.asarray(initializer(shape), dtype=np.float32))
    name = (initializer or 'glorot_uniform')
    if isinstance(name, dict):
        name = name.get('class_name', 'glorot_uniform')
    name = _to_snake(str(name))
    rng = _rng()
    fan_in, fan_out = _fans(shape)
    if name in ('zeros', 'zero'):
        a = np.zeros(shape)
    elif name in ('ones', 'one'):
        a = np.ones(shape)
    elif name in ('uniform', 'random_uniform'):
        a = rng.uniform(-0.05, 0.05, shape)
    elif name in ('normal', 'random_normal'):
        a = rng.normal(0.0, 0.05, shape)
    elif name == 'glorot_uniform':
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        a = rng.uniform(-lim, lim, shape)
    elif name == 'glorot_normal':
        a = rng.normal(0.0, math.sqrt(2.0 / (fan_in + fan_out)), shape)
    elif name == 'he_uniform':
        lim = math.sqrt(6.0 / fan_in)
        a = rng.uniform(-lim, lim, shape)
    elif name == 'he_normal':
        a = rng.normal(0.0, math.sqrt(2.0 / fan_in), shape)
    elif name == 'lecun_uniform':
        lim = math.sqrt(3.0 / fan_in)
        a = rng.uniform(-lim, lim, shape)
    else:
        raise ValueError(f'Unknown initializer: {initializer}')
    return torch.as_tensor(np.asarray(a, dtype=np.float32))


def get_activation(name):
    if name is None or name == 'linear':
        return None
    if callable(name):
        return name
    table = {'relu': torch.relu, 'tanh': torch.tanh, 'sigmoid': torch.sigmoid,
             'softmax': lambda t: torch.softmax(t, dim=-1), 'selu': torch.selu, 'elu': torch.nn.functional.elu,
             'softplus': torch.nn.functional.softplus, 'gelu': torch.nn.functional.gelu,
             'swish': torch.nn.functional.silu, 'softsign': torch.nn.functional.softsign, 'exponential': torch.exp}
    if name not in table:
        raise ValueError(f'Unknown activation: {name}')
    return table[name]


# ---------------------------------------------------------------------------------------------
# symbolic tensors and graph nodes
# ---------------------------------------------------------------------------------------------
class KTensor:
    """Symbolic tensor (batch dimension is None)."""

    def __init__(self, shape, dtype='float32', node=None, index=0, name=None):
        self.shape = tuple(shape)
        self.dtype = dtype
        self.node = node
        self.index = index
        self.name = name

    def __repr__(self):
        return f'<KTensor shape={self.shape} name={self.name}>'

    def __len__(self):
        raise TypeError('symbolic tensor has no len()')


class Node:
    def __init__(self, layer, inputs, outputs_struct):
        self.layer = layer
        self.inputs = inputs            # KTensor | list[KTensor]
        self.outputs = outputs_struct   # KTensor | list[KTensor]


def _flatten(x):
    if isinstance(x, (list, tuple)):
        out = []
        for e in x:
            out += _flatten(e)
        return out
    return [x]


def is_symbolic(x):
    return any(isinstance(t, KTensor) for t in _flatten(x))


def Input(shape, name=None, dtype='float32'):
    name = name or unique_name('input')
    t = KTensor((None,) + tuple(shape), dtype=dtype, name=name)
    t.is_input = True
    return t


def shape_of(x):
    if isinstance(x, (list, tuple)):
        return [shape_of(e) for e in x]
    return tuple(x.shape)


# ---------------------------------------------------------------------------------------------
# Layer base
# ---------------------------------------------------------------------------------------------
class Layer(nn.Module):
    def __init__(self, name=None, trainable=True, dtype=None, **kwargs):
        super().__init__()
        self._name = name or unique_name(_to_snake(self.__class__.__name__))
        self.built = False
        self.trainable = trainable
        self._outputs = None   # symbolic outputs of the (last) symbolic call

    @property
    def name(self):
        return self._name

    @property
    def output(self):
        return self._outputs

    # -- Keras hooks ---------------------------------------------------------------------------
    def build(self, input_shape):
        self.built = True

    def call(self, inputs, **kwargs):
        raise NotImplementedError

    def compute_output_shape(self, input_shape):
        return input_shape

    def get_config(self):
        return {'name': self.name, 'trainable': self.trainable}

    def add_weight(self, name=None, shape=None, initializer=None, trainable=True, dtype=None, **kwargs):
        p = nn.Parameter(initialize(shape, initializer), requires_grad=bool(trainable))
        self.register_parameter(name.replace('.', '_'), p)
        return p

    def _maybe_build(self, inputs):
        if not self.built:
            self.build(shape_of(inputs))
            self.built = True

    # -- call protocol ---------------------------------------------------------------------------
    def __call__(self, inputs, *args, **kwargs):
        if is_symbolic(inputs):
            self._maybe_build(inputs)
            out_shape = self.compute_output_shape(shape_of(inputs))
            node = Node(self, inputs, None)
            if isinstance(out_shape, list):
                outs = [KTensor(s, node=node, index=i, name=f'{self.name}:{i}') for i, s in enumerate(out_shape)]
            else:
                outs = KTensor(out_shape, node=node, index=0, name=self.name)
            node.outputs = outs
            self._outputs = outs
            return outs
        return super().__call__(inputs, *args, **kwargs)

    def forward(self, inputs, *args, **kwargs):
        self._maybe_build(inputs)
        return self.call(inputs, *args, **kwargs)

    # weights in Keras order/naming -> {name: ndarray}
    def get_weights_dict(self):
        return {k: v.detach().cpu().numpy() for k, v in list(self.named_parameters(recurse=False)) +
                list(self.named_buffers(recurse=False))}


# ---------------------------------------------------------------------------------------------
# built-in Keras layers used by the reference graph
# ---------------------------------------------------------------------------------------------
_VENDOR_GEMM_NOTED = set()


def note_vendor_gemm(who, x_shape, w_shape):
    """One log line per (layer, weight shape) when a product leaves the hand-written kernels (csrc/dense.hip takes K and N
    up to its LDS tile; beyond that the GEMM goes to the vendor library through torch): never silent."""
    key = (who, tuple(w_shape))
    if key in _VENDOR_GEMM_NOTED:
        return
    _VENDOR_GEMM_NOTED.add(key)
    import logging
    logging.getLogger('deeptables_amd').warning(
        '%s: [%s] x [%s] is outside csrc/dense.hip\'s tile -> vendor GEMM (torch.addmm / rocBLAS), not a hand-written kernel',
        who, ' x '.join(str(int(v)) for v in x_shape), ' x '.join(str(int(v)) for v in w_shape))


class Dense(Layer):
    """keras.layers.Dense: y = act(x @ kernel + bias), kernel [in, out] (glorot_uniform, zeros)."""

    @property
    def accepts_pending_norm(self):
        return self.units == 1 and self.activation_name in (None, 'linear')

    def __init__(self, units, activation=None, use_bias=True, kernel_initializer='glorot_uniform',
                 bias_initializer='zeros', kernel_regularizer=None, activity_regularizer=None, **kwargs):
        super().__init__(**kwargs)
        self.units = int(units)
        self.activation_name = activation if not callable(activation) else getattr(activation, '__name__', 'fn')
        self.activation = get_activation(activation)
        self.use_bias = use_bias
        self.kernel_initializer = kernel_initializer
        self.bias_initializer = bias_initializer

    def build(self, input_shape):
        self.kernel = self.add_weight('kernel', (input_shape[-1], self.units), self.kernel_initializer)
        self.bias = self.add_weight('bias', (self.units,), self.bias_initializer) if self.use_bias else None
        self.built = True

    def compute_output_shape(self, input_shape):
        return tuple(input_shape[:-1]) + (self.units,)

    def call(self, x, fused_activation=None, **kwargs):
        """fused_activation: an activation the model's execution plan folds into this layer (Model.__init__)."""
        act_name = self.activation_name if self.activation_name not in (None, 'linear') else fused_activation
        activation = self.activation if self.activation is not None else get_activation(fused_activation)
        lk = getattr(x, '_dt_bn_link', None)
        if lk is not None and lk.lazy:
            # the flattened output of an interacting layer whose BatchNormalization is pending (Model.__init__'s peephole):
            # a linear Dense(1) takes it as it is (ops.autoint_head), anything else gets the normalised tensor
            if self.units == 1 and act_name in (None, 'linear') and self.training and \
                    ops.autoint_head_supported(x, self.kernel, lk):
                return ops.autoint_head(x, self.kernel, self.bias)
            x = ops.autoint_materialize(x)
        fused_act = act_name if act_name in (None, 'linear', 'relu') else None
        if x.is_cuda and ops.dense_supported(x, self.kernel):
            # hand-written fp32 MFMA / GEMV kernels (csrc/dense.hip); relu and bias fused
            y = ops.dense(x, self.kernel, self.bias, fused_act)
            if activation is not None and fused_act is None:
                y = activation(y)
            return y
        lead = x.shape[:-1]                      # shapes outside the kernels' LDS tile: vendor GEMM
        note_vendor_gemm(f'Dense {self.name!r}', x.shape, self.kernel.shape)
        x2 = x.reshape(-1, x.shape[-1])
        y = torch.addmm(self.bias, x2, self.kernel) if self.bias is not None else x2 @ self.kernel
        if activation is not None:
            y = activation(y)
        return y.reshape(*lead, self.units)

    def get_config(self):
        c = super().get_config()
        c.update(units=self.units, activation=self.activation_name, use_bias=self.use_bias)
        return c


class Activation(Layer):
    def __init__(self, activation, **kwargs):
        super().__init__(**kwargs)
        self.activation_name = activation
        self.fn = get_activation(activation)

    def call(self, x, **kwargs):
        return x if self.fn is None else self.fn(x)


class Dropout(Layer):
    def __init__(self, rate, **kwargs):
        super().__init__(**kwargs)
        self.rate = float(rate)

    def call(self, x, **kwargs):
        if self.training and self.rate > 0:
            return torch.nn.functional.dropout(x, self.rate, True)
        return x


class SpatialDropout1D(Dropout):
    """On a [B,1,D] embedding (layers.py:878-880) this is element dropout with 1/(1-p) scaling."""

    def call(self, x, **kwargs):
        if self.training and self.rate > 0:
            mask = torch.nn.functional.dropout(torch.ones_like(x[:, :1, :]), self.rate, True)
            return x * mask
        return x


class Flatten(Layer):
    passes_pending_norm = True           # (Model.__init__'s peephole: a pending BatchNormalization travels through a reshape)

    def compute_output_shape(self, input_shape):
        return (input_shape[0], int(np.prod(input_shape[1:])))

    def call(self, x, **kwargs):
        y = x.reshape(x.shape[0], -1)
        lk = getattr(x, '_dt_bn_link', None)
        if lk is not None and lk.lazy:
            y._dt_bn_link = lk
        return y


def _packed_view(tensors, axis):
    """If `tensors` are, in order, all the per-column [B,1,D] views of one packed [B,F,D] embedding
    block, return the block reshaped for a concat along `axis` — the concat is then free."""
    first = getattr(tensors[0], '_dt_pack', None)
    if first is None:
        return None
    packed = first[0]
    if len(tensors) != packed.shape[1]:
        return None
    for i, t in enumerate(tensors):
        p = getattr(t, '_dt_pack', None)
        if p is None or p[0] is not packed or p[1] != i:
            return None
    nd = tensors[0].dim()
    ax = axis if axis >= 0 else nd + axis
    if ax == 1:
        return packed
    if ax == 2:
        return packed.reshape(packed.shape[0], 1, -1)
    return None


class Concatenate(Layer):
    def __init__(self, axis=-1, **kwargs):
        super().__init__(**kwargs)
        self.axis = axis

    def compute_output_shape(self, input_shape):
        shapes = list(input_shape)
        nd = len(shapes[0])
        ax = self.axis if self.axis >= 0 else nd + self.axis
        out = list(shapes[0])
        out[ax] = sum(s[ax] for s in shapes)
        return tuple(out)

    def call(self, xs, **kwargs):
        xs = list(xs)
        if len(xs) == 1:
            return xs[0]
        pv = _packed_view(xs, self.axis)
        if pv is not None:
            return pv
        return torch.cat(xs, dim=self.axis)


class Add(Layer):
    def compute_output_shape(self, input_shape):
        return tuple(input_shape[0])

    def call(self, xs, **kwargs):
        out = xs[0]
        for t in xs[1:]:
            out = out + t
        return out


class Lambda(Layer):
    """keras.layers.Lambda(fn, output_shape=callable(input_shape))."""

    def __init__(self, function, output_shape=None, **kwargs):
        super().__init__(**kwargs)
        self.function = function
        self._out_shape = output_shape

    def compute_output_shape(self, input_shape):
        if callable(self._out_shape):
            return self._out_shape(input_shape)
        if self._out_shape is not None:
            return (input_shape[0],) + tuple(self._out_shape)
        return input_shape

    def call(self, x, **kwargs):
        return self.function(x)


class ReduceSum(Layer):
    """keras.ops.sum(x, axis) as a layer (deepnets.py:51)."""

    def __init__(self, axis=-1, **kwargs):
        super().__init__(**kwargs)
        self.axis = axis

    def compute_output_shape(self, input_shape):
        s = list(input_shape)
        del s[self.axis]
        return tuple(s)

    def call(self, x, **kwargs):
        return torch.sum(x, dim=self.axis)


class BatchNormalization(Layer):
    """keras.layers.BatchNormalization over the last axis: momentum 0.99, epsilon 1e-3, biased batch
    variance, moving statistics updated with the biased variance (NOT torch.nn.BatchNorm defaults)."""

    def __init__(self, momentum=0.99, epsilon=1e-3, center=True, scale=True, **kwargs):
        super().__init__(**kwargs)
        self.momentum = momentum
        self.epsilon = epsilon
        self.center = center
        self.scale = scale

    def build(self, input_shape):
        C = input_shape[-1]
        self.gamma = self.add_weight('gamma', (C,), 'ones') if self.scale else None
        self.beta = self.add_weight('beta', (C,), 'zeros') if self.center else None
        self.register_buffer('moving_mean', torch.zeros(C))
        self.register_buffer('moving_variance', torch.ones(C))
        self.built = True

    def call(self, x, **kwargs):
        if self.training:
            return ops.batchnorm_train(x, self.gamma, self.beta, self.moving_mean, self.moving_variance,
                                       self.epsilon, self.momentum)
        return ops.batchnorm_infer(x, self.gamma, self.beta, self.moving_mean, self.moving_variance,
                                   self.epsilon)


# ---------------------------------------------------------------------------------------------
# Model: executes the recorded graph
# ---------------------------------------------------------------------------------------------
class Model(nn.Module):
    def __init__(self, inputs, outputs, name=None):
        super().__init__()
        self.inputs = _flatten(inputs)
        self.outputs_struct = outputs
        self.output_list = _flatten(outputs)
        self.name = name or unique_name('model')
        order, seen = [], set()

        def visit(t):
            node = t.node
            if node is None or id(node) in seen:
                return
            seen.add(id(node))
            for i in _flatten(node.inputs):
                visit(i)
            order.append(node)

        for t in self.output_list:
            visit(t)
        self.nodes = order
        self.layers_by_name = OrderedDict()
        for node in order:
            lname = node.layer.name
            if lname in self.layers_by_name and self.layers_by_name[lname] is not node.layer:
                raise ValueError(f'The name "{lname}" is used 2 times in the model. All layer names should be unique.')
            self.layers_by_name[lname] = node.layer
        self._layers = nn.ModuleList(list(self.layers_by_name.values()))
        self.input_names = [t.name for t in self.inputs]
        # peephole for THIS model's execution plan: Dense -> Activation('relu') where the Dense output feeds nothing
        # else and is not one of the model's outputs runs as one kernel (relu fused in the GEMM epilogue and in its
        # backward mask); the layers, their names and `apply(output_layers=[dense])` proxies are unaffected
        consumers = {}
        for node in order:
            for t in _flatten(node.inputs):
                consumers.setdefault(id(t), []).append(node)
        outs = {id(t) for t in self.output_list}
        self._fused_relu, self._passthrough = set(), set()
        for node in order:
            if isinstance(node.layer, Dense) and node.layer.activation_name in (None, 'linear') \
                    and not isinstance(node.outputs, list) and id(node.outputs) not in outs:
                cons = consumers.get(id(node.outputs), [])
                if len(cons) == 1 and isinstance(cons[0].layer, Activation) and cons[0].layer.activation_name == 'relu':
                    self._fused_relu.add(id(node))
                    self._passthrough.add(id(cons[0]))
        # second peephole: a layer that can hand over its output with the trailing BatchNormalization PENDING (the AutoInt
        # interacting layer) does so when that output has exactly one consumer and the consumer applies the normalisation while
        # it loads its input — the next interacting layer, or (through a Flatten that feeds nothing else) a linear Dense(1): the
        # normalised tensor is then never written (csrc/autoint.hip AiXn, k_autoint_head_*).  A consumer that cannot take the
        # pending form at run time (inference, unsupported shape) materialises it itself.
        self._defer_norm = set()
        for node in order:
            if not getattr(node.layer, 'can_defer_output_norm', False) or isinstance(node.outputs, list) or \
                    id(node.outputs) in outs:
                continue
            cons = consumers.get(id(node.outputs), [])
            if len(cons) == 1 and getattr(cons[0].layer, 'passes_pending_norm', False) and \
                    not isinstance(cons[0].outputs, list) and id(cons[0].outputs) not in outs and \
                    not isinstance(cons[0].inputs, (list, tuple)):
                cons = consumers.get(id(cons[0].outputs), [])
            if len(cons) == 1 and getattr(cons[0].layer, 'accepts_pending_norm', False) and \
                    not isinstance(cons[0].inputs, (list, tuple)) and id(cons[0]) not in self._fused_relu:
                self._defer_norm.add(id(node))

    @property
    def input(self):
        return self.inputs if len(self.inputs) > 1 else self.inputs[0]

    @property
    def layers(self):
        return list(self.layers_by_name.values())

    def get_layer(self, name):
        if name not in self.layers_by_name:
            raise ValueError(f'No such layer: {name}.')
        return self.layers_by_name[name]

    def forward(self, inputs):
        """inputs: list (same order as model inputs) or dict name -> tensor."""
        if isinstance(inputs, dict):
            vals = [inputs[n] for n in self.input_names]
        elif isinstance(inputs, (list, tuple)):
            vals = list(inputs)
        else:
            vals = [inputs]
        if len(vals) != len(self.inputs):
            raise ValueError(f'Model expects {len(self.inputs)} inputs but got {len(vals)}.')
        env = {id(t): v for t, v in zip(self.inputs, vals)}

        def fetch(x):
            if isinstance(x, (list, tuple)):
                return [fetch(e) for e in x]
            return env[id(x)]

        for node in self.nodes:
            if id(node) in self._passthrough:
                out = fetch(node.inputs)                      # its relu already ran inside the producing Dense
            elif id(node) in self._fused_relu:
                out = node.layer(fetch(node.inputs), fused_activation='relu')
            elif id(node) in self._defer_norm:
                out = node.layer(fetch(node.inputs), defer_bn=True)
            else:
                out = node.layer(fetch(node.inputs))
            if isinstance(node.outputs, list):
                for t, v in zip(node.outputs, out):
                    env[id(t)] = v
            else:
                env[id(node.outputs)] = out
        res = [env[id(t)] for t in self.output_list]
        return res if isinstance(self.outputs_struct, (list, tuple)) else res[0]

    def count_params(self):
        return sum(p.numel() for p in self.parameters())

    def get_weights_dict(self):
        out = OrderedDict()
        for lname, layer in self.layers_by_name.items():
            for k, v in list(layer.named_parameters()) + list(layer.named_buffers()):
                out[f'{lname}/{k}'] = v.detach().cpu().numpy()
        return out

    def summary(self, print_fn=print):
        print_fn(f'Model: {self.name}')
        for node in self.nodes:
            n = sum(p.numel() for p in node.layer.parameters())
            print_fn(f'  {node.layer.name:40s} {node.layer.__class__.__name__:24s} {shape_of(node.outputs)} params={n}')
        print_fn(f'Total params: {self.count_params()}')
