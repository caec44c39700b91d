# -*- coding:utf-8 -*-
"""Drop-in for `deeptables.models.deepnets` (deeptables/models/deepnets.py): the net-function
plugin API.  A net function has the signature of `linear` below (enforced by `register_nets`,
deepnets.py:496-502), receives the symbolic embeddings / dense tensors of the model under
construction and returns a tensor (or None to opt out, deepmodel.py:284).
"""
from inspect import signature

from . import layers
from ..functional import Dense, Concatenate, Flatten, BatchNormalization, Activation, Dropout, ReduceSum

WideDeep = ['linear', 'dnn_nets']
DeepFM = ['linear', 'fm_nets', 'dnn_nets']
xDeepFM = ['linear', 'cin_nets', 'dnn_nets']
AutoInt = ['autoint_nets']
DCN = ['dcn_nets']
FGCNN = ['fgcnn_dnn_nets']
FiBiNet = ['fibi_dnn_nets']
PNN = ['pnn_nets']
AFM = ['afm_nets']


def _concat_embeddings(embeddings, concat_layer_name):
    if embeddings is None or len(embeddings) == 0:
        return None
    if len(embeddings) == 1:
        return embeddings[0]
    return Concatenate(axis=1, name=concat_layer_name)(embeddings)


def linear(embeddings, flatten_emb_layer, dense_layer, concat_emb_dense, config, model_desc):
    """Linear(order-1) interactions (deepnets.py:43-66)."""
    x_emb = None
    concat_embeddings = _concat_embeddings(embeddings, 'concat_linear_embedding')
    if concat_embeddings is not None:
        x_emb = ReduceSum(axis=-1, name='linear_reduce_sum')(concat_embeddings)
    if x_emb is not None and dense_layer is not None:
        x = Concatenate(name='concat_linear_emb_dense')([x_emb, dense_layer])
    elif x_emb is not None:
        x = x_emb
    elif dense_layer is not None:
        x = dense_layer
    else:
        raise ValueError('No input layer exists.')
    input_shape = x.shape
    x = Dense(1, activation=None, use_bias=False, name='linear_logit')(x)
    model_desc.add_net('linear', input_shape, x.shape)
    return x


def cin_nets(embeddings, flatten_emb_layer, dense_layer, concat_emb_dense, config, model_desc):
    """Compressed Interaction Network (deepnets.py:69-81)."""
    cin_concat = _concat_embeddings(embeddings, 'concat_cin_embedding')
    if cin_concat is None:
        model_desc.add_net('cin', (None), (None))
        return None
    cin_output = layers.CIN(params=config.cin_params)(cin_concat)
    model_desc.add_net('cin', cin_concat.shape, cin_output.shape)
    return cin_output


def fm_nets(embeddings, flatten_emb_layer, dense_layer, concat_emb_dense, config, model_desc):
    """FM pairwise (order-2) interactions (deepnets.py:84-96)."""
    concat_embeddings_layer = _concat_embeddings(embeddings, 'concat_fm_embedding')
    if concat_embeddings_layer is None:
        model_desc.add_net('fm', (None), (None))
        return None
    fm_output = layers.FM(name='fm_layer')(concat_embeddings_layer)
    model_desc.add_net('fm', concat_embeddings_layer.shape, fm_output.shape)
    return fm_output


def afm_nets(embeddings, flatten_emb_layer, dense_layer, concat_emb_dense, config, model_desc):
    """Attentional FM (deepnets.py:99-108)."""
    if embeddings is None or len(embeddings) < 2:
        return None
    afm_output = layers.AFM(params=config.afm_params, name='afm_layer')(embeddings)
    model_desc.add_net('afm', f'list({len(embeddings)})', afm_output.shape)
    return afm_output


def opnn_nets(embeddings, flatten_emb_layer, dense_layer, concat_emb_dense, config, model_desc):
    """OuterProduct + DNN (deepnets.py:111-125)."""
    if embeddings is None or len(embeddings) < 2:
        return None
    op = layers.OuterProduct(config.pnn_params, name='outer_product_layer')(embeddings)
    model_desc.add_net('opnn-outer_product', f'list({len(embeddings)})', op.shape)
    concat_all = Concatenate(name='concat_opnn_all')([op, concat_emb_dense])
    x_dnn = dnn(concat_all, config.dnn_params, cellname='opnn')
    model_desc.add_net('opnn-dnn', concat_all.shape, x_dnn.shape)
    return x_dnn


def ipnn_nets(embeddings, flatten_emb_layer, dense_layer, concat_emb_dense, config, model_desc):
    """InnerProduct + DNN (deepnets.py:128-141)."""
    if embeddings is None or len(embeddings) < 2:
        return None
    ip = layers.InnerProduct(name='inner_product_layer')(embeddings)
    model_desc.add_net('ipnn-inner_product', f'list({len(embeddings)})', ip.shape)
    concat_all = Concatenate(name='concat_ipnn_all')([ip, concat_emb_dense])
    x_dnn = dnn(concat_all, config.dnn_params, cellname='ipnn')
    model_desc.add_net('ipnn-dnn', concat_all.shape, x_dnn.shape)
    return x_dnn


def pnn_nets(embeddings, flatten_emb_layer, dense_layer, concat_emb_dense, config, model_desc):
    """Inner product ++ outer product + DNN (deepnets.py:144-160)."""
    if embeddings is None or len(embeddings) < 2:
        return None
    ip = layers.InnerProduct(name='pnn_inner_product_layer')(embeddings)
    model_desc.add_net('pnn-inner_product', f'list({len(embeddings)})', ip.shape)
    op = layers.OuterProduct(params=config.pnn_params, name='pnn_outer_product_layer')(embeddings)
    model_desc.add_net('pnn-outer_product', f'list({len(embeddings)})', op.shape)
    concat_all = Concatenate(name='concat_pnn_all')([ip, op, concat_emb_dense])
    x_dnn = dnn(concat_all, config.dnn_params, cellname='pnn')
    model_desc.add_net('pnn-dnn', concat_all.shape, x_dnn.shape)
    return x_dnn


def dnn_nets(embeddings, flatten_emb_layer, dense_layer, concat_emb_dense, config, model_desc):
    """MLP (deepnets.py:163-169)."""
    x_dnn = dnn(concat_emb_dense, config.dnn_params)
    model_desc.add_net('dnn', concat_emb_dense.shape, x_dnn.shape)
    return x_dnn


def cross_nets(embeddings, flatten_emb_layer, dense_layer, concat_emb_dense, config, model_desc):
    """Cross network (deepnets.py:172-178)."""
    cross = layers.Cross(params=config.cross_params, name='cross_layer')(concat_emb_dense)
    model_desc.add_net('cross', concat_emb_dense.shape, cross.shape)
    return cross


def cross_dnn_nets(embeddings, flatten_emb_layer, dense_layer, concat_emb_dense, config, model_desc):
    """Cross -> DNN (deepnets.py:181-191)."""
    x = concat_emb_dense
    cross = layers.Cross(params=config.cross_params, name='cross_dnn_layer')(x)
    model_desc.add_net('cross_dnn-cross', x.shape, cross.shape)
    x_dnn = dnn(cross, config.dnn_params, cellname='cross_dnn')
    model_desc.add_net('cross_dnn-dnn', cross.shape, x_dnn.shape)
    return x_dnn


def dcn_nets(embeddings, flatten_emb_layer, dense_layer, concat_emb_dense, config, model_desc):
    """Cross || DNN, concatenated (deepnets.py:194-207)."""
    x = concat_emb_dense
    cross_out = layers.Cross(params=config.cross_params, name='dcn_cross_layer')(x)
    model_desc.add_net('dcn-widecross', x.shape, cross_out.shape)
    dnn_out = dnn(x, config.dnn_params, cellname='dcn')
    model_desc.add_net('dcn-dnn2', x.shape, dnn_out.shape)
    stack_out = Concatenate(name='concat_cross_dnn')([cross_out, dnn_out])
    model_desc.add_net('dcn', x.shape, stack_out.shape)
    return stack_out


def autoint_nets(embeddings, flatten_emb_layer, dense_layer, concat_emb_dense, config, model_desc):
    """AutoInt: stacked multi-head field self-attention (deepnets.py:210-224)."""
    concat_embeddings_layer = _concat_embeddings(embeddings, 'concat_autoint_embedding')
    if concat_embeddings_layer is None:
        model_desc.add_net('autoint', (None), (None))
        return None
    output = concat_embeddings_layer
    for i in range(config.autoint_params['num_attention']):
        output = layers.MultiheadAttention(params=config.autoint_params)(output)
    output = Flatten()(output)
    model_desc.add_net('autoint', concat_embeddings_layer.shape, output.shape)
    return output


def _not_yet(name):
    def fn(embeddings, flatten_emb_layer, dense_layer, concat_emb_dense, config, model_desc):
        raise NotImplementedError(f'net "{name}" needs layer types outside this round\'s accelerated hot path '
                                  f'(SURVEY §8 f3: FGCNN / SENET / BilinearInteraction).')
    fn.__name__ = name
    return fn


fg_nets = _not_yet('fg_nets')
fgcnn_cin_nets = _not_yet('fgcnn_cin_nets')
fgcnn_fm_nets = _not_yet('fgcnn_fm_nets')
fgcnn_afm_nets = _not_yet('fgcnn_afm_nets')
fgcnn_ipnn_nets = _not_yet('fgcnn_ipnn_nets')
fgcnn_dnn_nets = _not_yet('fgcnn_dnn_nets')
fibi_nets = _not_yet('fibi_nets')
fibi_dnn_nets = _not_yet('fibi_dnn_nets')


def dnn(x, params, cellname='dnn'):
    """[Dense -> (BN) -> Activation -> (Dropout)]* (deepnets.py:401-427)."""
    custom_dnn_fn = params.get('custom_dnn_fn')
    if custom_dnn_fn is not None:
        return custom_dnn_fn(x, params, cellname + '_custom')
    hidden_units = params.get('hidden_units', ((128, 0, True), (64, 0, False)))
    activation = params.get('activation', 'relu')
    kernel_initializer = params.get('kernel_initializer', 'he_uniform')
    if len(hidden_units) <= 0:
        raise ValueError(
            '[hidden_units] must be a list of tuple([units],[dropout_rate],[use_bn]) and at least one tuple.')
    index = 1
    for units, dropout, batch_norm in hidden_units:
        x = Dense(units, use_bias=not batch_norm, name=f'{cellname}_dense_{index}',
                  kernel_initializer=kernel_initializer)(x)
        if batch_norm:
            x = BatchNormalization(name=f'{cellname}_bn_{index}')(x)
        x = Activation(activation=activation, name=f'{cellname}_activation_{index}')(x)
        if dropout > 0:
            x = Dropout(dropout, name=f'{cellname}_dropout_{index}')(x)
        index += 1
    return x


def custom_dnn_D_A_D_B(x, params, cellname='dnn_D_A_D_B'):
    """Dense(act) -> Dropout -> BN cell order (deepnets.py:430-452)."""
    hidden_units = params.get('hidden_units', ((128, 0, True), (64, 0, False)))
    activation = params.get('activation', 'relu')
    kernel_initializer = params.get('kernel_initializer', 'he_uniform')
    if len(hidden_units) <= 0:
        raise ValueError(
            '[hidden_units] must be a list of tuple([units],[dropout_rate],[use_bn]) and at least one tuple.')
    index = 1
    for units, dropout, batch_norm in hidden_units:
        x = Dense(units, activation=activation, kernel_initializer=kernel_initializer,
                  name=f'{cellname}_dense_{index}')(x)
        if dropout > 0:
            x = Dropout(dropout, name=f'{cellname}_dropout_{index}')(x)
        if batch_norm:
            x = BatchNormalization(name=f'{cellname}_bn_{index}')(x)
        index += 1
    return x


def get(identifier):
    """Name or callable -> net function (deepnets.py:455-478)."""
    if identifier is None:
        raise ValueError(f'identifier can not be none.')
    if isinstance(identifier, str):
        nets_fn = custom_nets.get(identifier)
        if nets_fn is not None:
            return nets_fn
        fn = globals().get(identifier)
        if fn is None or not callable(fn):
            raise ValueError(f'Unknown nets function: {identifier}')
        return fn
    elif callable(identifier):
        register_nets(identifier)
        return identifier
    else:
        raise TypeError(f'Could not interpret nets function identifier: {repr(identifier)}')


custom_nets = {}


def get_nets(nets):
    """Names of the nets, callables registered on the fly (deepnets.py:484-493).  The reference
    dedups through set() which makes the order hash-dependent; here duplicates
    are dropped but the caller's order is kept, which fixes the Add order of the logits."""
    str_nets = []
    for net in nets:
        name = net if isinstance(net, str) else register_nets(net)
        if name not in str_nets:
            str_nets.append(name)
    return str_nets


def register_nets(nets_fn):
    if not callable(nets_fn):
        raise ValueError('nets_fn must be a valid callable function.')
    if signature(nets_fn) != signature(linear):
        raise ValueError(f'Signature of nets_fn is invalid, except {signature(linear)}  but {signature(nets_fn)}')
    custom_nets[nets_fn.__name__] = nets_fn
    return nets_fn.__name__
