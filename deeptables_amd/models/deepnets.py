# -*- coding:utf-8 -*-
"""Drop-in for `deeptables.models.deepnets` (DataCanvasIO/DeepTables, Apache-2.0; reference file
deeptables/models/deepnets.py): the net-function plugin API.

What is kept from the reference is its INTERFACE, because models, checkpoints and user plugins depend on it: the names of
the net functions and presets, the six-argument plugin signature (`register_nets` compares against `linear`,
deepnets.py:496-502), the Keras layer names (1:1 weight mapping with a Keras checkpoint) and the keys / order of the
`model_desc.add_net` records.  How the graphs are put together is this repo's: every net body works on a `_Wiring`
object (the six plugin arguments + the recurring moves: stack the field embeddings, describe a net, run the DNN tower)
and the FGCNN / product families are generated from tables instead of being written out per net.
"""
from inspect import signature

from . import layers
from ..functional import Dense, Concatenate, Flatten, BatchNormalization, Activation, Dropout, ReduceSum, Layer
from ..utils import counter

# presets (deepnets.py:14-22)
WideDeep = ['linear', 'dnn_nets']
DeepFM = ['linear', 'fm_nets', 'dnn_nets']
xDeepFM = ['linear', 'cin_nets', 'dnn_nets']
AutoInt = ['autoint_nets']
DCN = ['dcn_nets']
FGCNN = ['fgcnn_dnn_nets']
FiBiNet = ['fibi_dnn_nets']
PNN = ['pnn_nets']
AFM = ['afm_nets']

_HIDDEN_DEFAULT = ((128, 0, True), (64, 0, False))          # deepnets.py:406 (ModelConfig always overrides it, config.py:90-93)
_HIDDEN_ERROR = '[hidden_units] must be a list of tuple([units],[dropout_rate],[use_bn]) and at least one tuple.'


class _Wiring:
    """The six arguments of a net function and the moves every net repeats."""

    __slots__ = ('embeddings', 'flat', 'dense', 'xn', 'config', 'desc')

    def __init__(self, embeddings, flat, dense, xn, config, desc):
        self.embeddings, self.flat, self.dense, self.xn, self.config, self.desc = embeddings, flat, dense, xn, config, desc

    def fields(self, concat_name):
        """[B,F,D] stack of the field embeddings (None without categorical columns; one column: itself)"""
        e = self.embeddings
        if not e:
            return None
        return e[0] if len(e) == 1 else Concatenate(axis=1, name=concat_name)(e)

    def pairs(self):
        """the embedding list for the pairwise layers, None below two fields"""
        e = self.embeddings
        return e if e is not None and len(e) >= 2 else None

    def note(self, key, src, out):
        """model_desc record; src / out: a tensor (its shape is recorded), a string, or None"""
        self.desc.add_net(key, getattr(src, 'shape', src), getattr(out, 'shape', out))
        return out

    def absent(self, key):
        self.desc.add_net(key, (None), (None))
        return None

    def tower(self, x, cell, key):
        out = dnn(x, self.config.dnn_params, cellname=cell)
        self.desc.add_net(key, x.shape, out.shape)
        return out

    def n_fields(self):
        return f'list({len(self.embeddings)})'


def _plugin(body):
    """body(_Wiring) -> a net function with the reference's plugin signature and the body's name"""
    def net(embeddings, flatten_emb_layer, dense_layer, concat_emb_dense, config, model_desc):
        return body(_Wiring(embeddings, flatten_emb_layer, dense_layer, concat_emb_dense, config, model_desc))
    net.__name__ = net.__qualname__ = body.__name__
    net.__doc__ = body.__doc__
    return net


# ---- order-1 / order-2 / attention / compressed interactions --------------------------------------------------------
@_plugin
def linear(w):
    """Linear (order-1) interactions: Dense(1, no bias) over [sum_D(embeddings), dense] (deepnets.py:43-66)."""
    stack = w.fields('concat_linear_embedding')
    parts = []
    if stack is not None:
        parts.append(ReduceSum(axis=-1, name='linear_reduce_sum')(stack))
    if w.dense is not None:
        parts.append(w.dense)
    if not parts:
        raise ValueError('No input layer exists.')
    x = parts[0] if len(parts) == 1 else Concatenate(name='concat_linear_emb_dense')(parts)
    return w.note('linear', x, Dense(1, activation=None, use_bias=False, name='linear_logit')(x))


def _on_field_stack(fn_name, concat_name, key, make_layer, doc):
    """a net that is ONE layer over the stacked field embeddings (cin_nets, fm_nets)"""
    def body(w):
        stack = w.fields(concat_name)
        return w.absent(key) if stack is None else w.note(key, stack, make_layer(w.config)(stack))
    body.__name__, body.__doc__ = fn_name, doc
    return _plugin(body)


cin_nets = _on_field_stack('cin_nets', 'concat_cin_embedding', 'cin', lambda c: layers.CIN(params=c.cin_params),
                           'Compressed Interaction Network (deepnets.py:69-81; layers.py:638-734).')
fm_nets = _on_field_stack('fm_nets', 'concat_fm_embedding', 'fm', lambda c: layers.FM(name='fm_layer'),
                          'FM pairwise (order-2) interactions (deepnets.py:84-96; layers.py:53-62).')


@_plugin
def afm_nets(w):
    """Attentional FM (deepnets.py:99-108)."""
    e = w.pairs()
    if e is None:
        return None
    return w.note('afm', w.n_fields(), layers.AFM(params=w.config.afm_params, name='afm_layer')(e))


@_plugin
def autoint_nets(w):
    """AutoInt: `num_attention` stacked multi-head field self-attention layers, flattened (deepnets.py:210-224)."""
    stack = w.fields('concat_autoint_embedding')
    if stack is None:
        return w.absent('autoint')
    x = stack
    for _ in range(w.config.autoint_params['num_attention']):
        x = layers.MultiheadAttention(params=w.config.autoint_params)(x)
    return w.note('autoint', stack, Flatten()(x))


# ---- product networks: <products of the embedding pairs> ++ BN(concat) -> DNN -----------------------------------------
def _product_net(fn_name, cell, products, doc):
    """products: [(record key, layer factory(config))] applied to the embedding list, in order"""
    def body(w):
        e = w.pairs()
        if e is None:
            return None
        outs = [w.note(key, w.n_fields(), make(w.config)(e)) for key, make in products]
        return w.tower(Concatenate(name=f'concat_{cell}_all')(outs + [w.xn]), cell, f'{cell}-dnn')
    body.__name__, body.__doc__ = fn_name, doc
    return _plugin(body)


opnn_nets = _product_net('opnn_nets', 'opnn',
                         [('opnn-outer_product', lambda c: layers.OuterProduct(c.pnn_params, name='outer_product_layer'))],
                         'OuterProduct + DNN (deepnets.py:111-125).')
ipnn_nets = _product_net('ipnn_nets', 'ipnn',
                         [('ipnn-inner_product', lambda c: layers.InnerProduct(name='inner_product_layer'))],
                         'InnerProduct + DNN (deepnets.py:128-141).')
pnn_nets = _product_net('pnn_nets', 'pnn',
                        [('pnn-inner_product', lambda c: layers.InnerProduct(name='pnn_inner_product_layer')),
                         ('pnn-outer_product', lambda c: layers.OuterProduct(params=c.pnn_params, name='pnn_outer_product_layer'))],
                        'Inner product ++ outer product + DNN (deepnets.py:144-160).')


# ---- towers on BN(concat(embeddings, dense)) -------------------------------------------------------------------------
@_plugin
def dnn_nets(w):
    """MLP (deepnets.py:163-169)."""
    return w.tower(w.xn, 'dnn', 'dnn')


@_plugin
def cross_nets(w):
    """Cross network (deepnets.py:172-178; layers.py:417-436)."""
    return w.note('cross', w.xn, layers.Cross(params=w.config.cross_params, name='cross_layer')(w.xn))


@_plugin
def cross_dnn_nets(w):
    """Cross -> DNN (deepnets.py:181-191)."""
    crossed = w.note('cross_dnn-cross', w.xn, layers.Cross(params=w.config.cross_params, name='cross_dnn_layer')(w.xn))
    return w.tower(crossed, 'cross_dnn', 'cross_dnn-dnn')


@_plugin
def dcn_nets(w):
    """Cross || DNN on the same input, concatenated (deepnets.py:194-207)."""
    wide = w.note('dcn-widecross', w.xn, layers.Cross(params=w.config.cross_params, name='dcn_cross_layer')(w.xn))
    deep = w.tower(w.xn, 'dcn', 'dcn-dnn2')
    return w.note('dcn', w.xn, Concatenate(name='concat_cross_dnn')([wide, deep]))


# ---- FGCNN -----------------------------------------------------------------------------------------------------------
class _ExpandDims(Layer):
    """keras.ops.expand_dims(x, -1) (deepnets.py:245)."""

    def compute_output_shape(self, input_shape):
        return tuple(input_shape) + (1,)

    def call(self, x, **kwargs):
        return x.unsqueeze(-1)


class _SplitFields(Layer):
    """keras.ops.split(x, x.shape[1], axis=1) -> list of [B,1,D] (deepnets.py:301,315).  The pieces carry the
    packed-view tag so the pairwise layers downstream re-stack them for free."""

    def compute_output_shape(self, input_shape):
        return [(input_shape[0], 1, input_shape[2]) for _ in range(int(input_shape[1]))]

    def call(self, x, **kwargs):
        x = x.contiguous()
        outs = []
        for i in range(x.shape[1]):
            v = x[:, i:i + 1, :]
            v._dt_pack = (x, i)
            outs.append(v)
        return outs


_FG_DEFAULTS = (('fg_filters', (14, 16)), ('fg_heights', (7, 7)), ('fg_pool_heights', (2, 2)), ('fg_new_feat_filters', (2, 2)))


def _feature_generation(w):
    """FGCNN feature generation (deepnets.py:227-261): stacked Conv / MaxPool / recombine blocks; the generated features of
    every block ++ the raw embeddings along the field axis.  None (and an empty record) without categorical columns."""
    raw = w.fields('concat_fgcnn_embedding_%s' % counter.next_num('concat_fgcnn_embedding'))
    if raw is None:
        return w.absent('fgcnn')
    x = _ExpandDims()(raw)
    generated = []
    for filters, height, pool, new_filters in zip(*(w.config.fgcnn_params.get(k, d) for k, d in _FG_DEFAULTS)):
        x, feats = layers.FGCNN(filters=filters, kernel_height=height, pool_height=pool, new_filters=new_filters)(x)
        generated.append(feats)
    return w.note('fg', raw, Concatenate(axis=1)(generated + [raw]))


fg_nets = _plugin(_feature_generation)
fg_nets.__name__ = fg_nets.__qualname__ = 'fg_nets'


def _fg_flat_with_dense(w, fg, extra=()):
    parts = [Flatten()(fg)] + list(extra)
    if w.dense is not None:
        parts.append(w.dense)
    return parts[0] if len(parts) == 1 else Concatenate()(parts)


def _fg_head_cin(w, fg):
    return layers.CIN(params=w.config.cin_params)(fg)


def _fg_head_fm(w, fg):
    return layers.FM(name='fm_fgcnn_layer')(fg)


def _fg_head_afm(w, fg):
    return layers.AFM(params=w.config.afm_params)(_SplitFields()(fg))


def _fg_head_ipnn(w, fg):
    ip = layers.InnerProduct()(_SplitFields()(fg))
    return dnn(_fg_flat_with_dense(w, fg, [ip]), w.config.dnn_params, cellname='fgcnn_ipnn')


def _fg_head_dnn(w, fg):
    return dnn(_fg_flat_with_dense(w, fg), w.config.dnn_params, cellname='fgcnn_dnn')


def _fg_classifier(fn_name, key, head, doc):
    """FGCNN's generated features -> `head` as the deep classifier"""
    def body(w):
        fg = _feature_generation(w)
        return None if fg is None else w.note(key, fg, head(w, fg))
    body.__name__, body.__doc__ = fn_name, doc
    return _plugin(body)


fgcnn_cin_nets = _fg_classifier('fgcnn_cin_nets', 'fgcnn-cin', _fg_head_cin, 'FGCNN with CIN as deep classifier (deepnets.py:264-275).')
fgcnn_fm_nets = _fg_classifier('fgcnn_fm_nets', 'fgcnn-fm', _fg_head_fm, 'FGCNN with FM as deep classifier (deepnets.py:278-289).')
fgcnn_afm_nets = _fg_classifier('fgcnn_afm_nets', 'fgcnn-afm', _fg_head_afm, 'FGCNN with AFM as deep classifier (deepnets.py:292-303).')
fgcnn_ipnn_nets = _fg_classifier('fgcnn_ipnn_nets', 'fgcnn-ipnn', _fg_head_ipnn,
                                 'FGCNN with IPNN as deep classifier (deepnets.py:306-323).')
# (the reference records this one under 'fgcnn-ipnn' too, deepnets.py:340)
fgcnn_dnn_nets = _fg_classifier('fgcnn_dnn_nets', 'fgcnn-ipnn', _fg_head_dnn, 'FGCNN with DNN as deep classifier (deepnets.py:326-341).')


# ---- FiBiNet ---------------------------------------------------------------------------------------------------------
def _fibi(w):
    """SENET re-weighted embeddings and the raw embeddings each through a bilinear interaction; the two [B,P,D] blocks
    concatenated along the pair axis (deepnets.py:344-371)."""
    n = counter.next_num('senet_layer')
    raw = w.fields(f'concat_senet_embedding_{n}')
    if raw is None:
        return w.absent('fibi')
    p = w.config.fibinet_params
    kind = p.get('bilinear_type', 'field_interaction')
    squeezed = layers.SENET(pooling_op=p.get('senet_pooling_op', 'mean'), reduction_ratio=p.get('senet_reduction_ratio', 3),
                            name=f'senet_layer_{n}')(raw)
    both = [layers.BilinearInteraction(bilinear_type=kind, name=f'senet_bilinear_layer_{n}')(squeezed),
            layers.BilinearInteraction(bilinear_type=kind, name=f'embedding_bilinear_layer_{n}')(raw)]
    return w.note('fibi', raw, Concatenate(axis=1, name=f'concat_bilinear_{n}')(both))


fibi_nets = _plugin(_fibi)
fibi_nets.__name__ = fibi_nets.__qualname__ = 'fibi_nets'


@_plugin
def fibi_dnn_nets(w):
    """FiBiNet with DNN as deep classifier (deepnets.py:374-386; like the reference it needs a dense input)."""
    if w.pairs() is None:
        return None
    fibi = _fibi(w)
    x = Concatenate(name='concat_bilinear_dense')([Flatten(name='flatten_fibi_output')(fibi), w.dense])
    out = dnn(x, w.config.dnn_params, cellname='fibi_dnn')
    return w.note('fibi-dnn', fibi, out)


# ---- the DNN tower ---------------------------------------------------------------------------------------------------
def _tower_cells(params):
    cells = params.get('hidden_units', _HIDDEN_DEFAULT)
    if len(cells) <= 0:
        raise ValueError(_HIDDEN_ERROR)
    return cells, params.get('activation', 'relu'), params.get('kernel_initializer', 'he_uniform')


def dnn(x, params, cellname='dnn'):
    """[Dense -> (BN) -> Activation -> (Dropout)]* (deepnets.py:401-427); params['custom_dnn_fn'] replaces the cell."""
    custom = params.get('custom_dnn_fn')
    if custom is not None:
        return custom(x, params, cellname + '_custom')
    cells, activation, init = _tower_cells(params)
    for i, (units, rate, use_bn) in enumerate(cells, start=1):
        x = Dense(units, use_bias=not use_bn, name=f'{cellname}_dense_{i}', kernel_initializer=init)(x)
        if use_bn:
            x = BatchNormalization(name=f'{cellname}_bn_{i}')(x)
        x = Activation(activation=activation, name=f'{cellname}_activation_{i}')(x)
        if rate > 0:
            x = Dropout(rate, name=f'{cellname}_dropout_{i}')(x)
    return x


def custom_dnn_D_A_D_B(x, params, cellname='dnn_D_A_D_B'):
    """the alternative cell order Dense(activation) -> (Dropout) -> (BN) (deepnets.py:430-452)."""
    cells, activation, init = _tower_cells(params)
    for i, (units, rate, use_bn) in enumerate(cells, start=1):
        x = Dense(units, activation=activation, kernel_initializer=init, name=f'{cellname}_dense_{i}')(x)
        if rate > 0:
            x = Dropout(rate, name=f'{cellname}_dropout_{i}')(x)
        if use_bn:
            x = BatchNormalization(name=f'{cellname}_bn_{i}')(x)
    return x


# ---- registry (deepnets.py:455-502) ----------------------------------------------------------------------------------
custom_nets = {}


def register_nets(nets_fn):
    """a user's net function joins the registry under its __name__; its signature must be `linear`'s"""
    if not callable(nets_fn):
        raise ValueError('nets_fn must be a valid callable function.')
    if signature(nets_fn) != signature(linear):
        raise ValueError(f'Signature of nets_fn is invalid, except {signature(linear)}  but {signature(nets_fn)}')
    custom_nets[nets_fn.__name__] = nets_fn
    return nets_fn.__name__


def get(identifier):
    """name or callable -> net function (registered nets shadow the built-in ones)"""
    if identifier is None:
        raise ValueError('identifier can not be none.')
    if callable(identifier):
        register_nets(identifier)
        return identifier
    if not isinstance(identifier, str):
        raise TypeError(f'Could not interpret nets function identifier: {repr(identifier)}')
    fn = custom_nets.get(identifier) or globals().get(identifier)
    if fn is None or not callable(fn):
        raise ValueError(f'Unknown nets function: {identifier}')
    return fn


def get_nets(nets):
    """names of the nets, callables registered on the fly (deepnets.py:484-493).  The reference dedups through set(),
    which makes the order hash-dependent; here duplicates are dropped but the caller's order is
    kept, which fixes the Add order of the logits."""
    names = []
    for net in nets:
        name = net if isinstance(net, str) else register_nets(net)
        if name not in names:
            names.append(name)
    return names
