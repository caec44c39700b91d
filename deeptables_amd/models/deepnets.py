# -*- coding:utf-8 -*-
"""Drop-in for `deeptables.models.deepnets` (deeptables/models/deepnets.py): the net-function
plugin API.  A net function has the signature of `linear` below (enforced by `register_nets`,
deepnets.py:496-502), receives the symbolic embeddings / dense tensors of the model under
construction and returns a tensor (or None to opt out, deepmodel.py:284).
"""
from inspect import signature

from . import layers
from ..functional import Dense, Concatenate, Flatten, BatchNormalization, Activation, Dropout, ReduceSum, Layer
from ..utils import counter

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


def fg_nets(embeddings, flatten_emb_layer, dense_layer, concat_emb_dense, config, model_desc):
    """FGCNN feature generation: stacked Conv/MaxPool/recombine blocks; output = new features ++ raw
    embeddings along the field axis (deepnets.py:227-261)."""
    fgcnn_emb_concat_index = counter.next_num('concat_fgcnn_embedding')
    fgcnn_emb_concat = _concat_embeddings(embeddings, f'concat_fgcnn_embedding_{fgcnn_emb_concat_index}')
    if fgcnn_emb_concat is None:
        model_desc.add_net('fgcnn', (None), (None))
        return None
    fg_inputs = _ExpandDims()(fgcnn_emb_concat)
    fg_filters = config.fgcnn_params.get('fg_filters', (14, 16))
    fg_heights = config.fgcnn_params.get('fg_heights', (7, 7))
    fg_pool_heights = config.fgcnn_params.get('fg_pool_heights', (2, 2))
    fg_new_feat_filters = config.fgcnn_params.get('fg_new_feat_filters', (2, 2))
    new_features = list()
    for filters, width, pool, new_filters in zip(fg_filters, fg_heights, fg_pool_heights, fg_new_feat_filters):
        fg_inputs, new_feats = layers.FGCNN(filters=filters, kernel_height=width, pool_height=pool,
                                            new_filters=new_filters)(fg_inputs)
        new_features.append(new_feats)
    concat_all_features = Concatenate(axis=1)(new_features + [fgcnn_emb_concat])
    model_desc.add_net('fg', fgcnn_emb_concat.shape, concat_all_features.shape)
    return concat_all_features


def fgcnn_cin_nets(embeddings, flatten_emb_layer, dense_layer, concat_emb_dense, config, model_desc):
    """FGCNN with CIN as deep classifier (deepnets.py:264-275)."""
    fg_output = fg_nets(embeddings, flatten_emb_layer, dense_layer, concat_emb_dense, config, model_desc)
    if fg_output is None:
        return None
    cin_output = layers.CIN(params=config.cin_params)(fg_output)
    model_desc.add_net('fgcnn-cin', fg_output.shape, cin_output.shape)
    return cin_output


def fgcnn_fm_nets(embeddings, flatten_emb_layer, dense_layer, concat_emb_dense, config, model_desc):
    """FGCNN with FM as deep classifier (deepnets.py:278-289)."""
    fg_output = fg_nets(embeddings, flatten_emb_layer, dense_layer, concat_emb_dense, config, model_desc)
    if fg_output is None:
        return None
    fm_output = layers.FM(name='fm_fgcnn_layer')(fg_output)
    model_desc.add_net('fgcnn-fm', fg_output.shape, fm_output.shape)
    return fm_output


def fgcnn_afm_nets(embeddings, flatten_emb_layer, dense_layer, concat_emb_dense, config, model_desc):
    """FGCNN with AFM as deep classifier (deepnets.py:292-303)."""
    fg_output = fg_nets(embeddings, flatten_emb_layer, dense_layer, concat_emb_dense, config, model_desc)
    if fg_output is None:
        return None
    split_features = _SplitFields()(fg_output)
    afm_output = layers.AFM(params=config.afm_params)(split_features)
    model_desc.add_net('fgcnn-afm', fg_output.shape, afm_output.shape)
    return afm_output


def fgcnn_ipnn_nets(embeddings, flatten_emb_layer, dense_layer, concat_emb_dense, config, model_desc):
    """FGCNN with IPNN as deep classifier (deepnets.py:306-323)."""
    fg_output = fg_nets(embeddings, flatten_emb_layer, dense_layer, concat_emb_dense, config, model_desc)
    if fg_output is None:
        return None
    split_features = _SplitFields()(fg_output)
    inner_product = layers.InnerProduct()(split_features)
    dnn_input_layers = [Flatten()(fg_output), inner_product]
    if dense_layer is not None:
        dnn_input_layers.append(dense_layer)
    dnn_input = Concatenate()(dnn_input_layers)
    dnn_out = dnn(dnn_input, config.dnn_params, cellname='fgcnn_ipnn')
    model_desc.add_net('fgcnn-ipnn', fg_output.shape, dnn_out.shape)
    return dnn_out


def fgcnn_dnn_nets(embeddings, flatten_emb_layer, dense_layer, concat_emb_dense, config, model_desc):
    """FGCNN with DNN as deep classifier (deepnets.py:326-341)."""
    fg_output = fg_nets(embeddings, flatten_emb_layer, dense_layer, concat_emb_dense, config, model_desc)
    if fg_output is None:
        return None
    if dense_layer is not None:
        dnn_input = Concatenate()([Flatten()(fg_output), dense_layer])
    else:
        dnn_input = Flatten()(fg_output)
    dnn_out = dnn(dnn_input, config.dnn_params, cellname='fgcnn_dnn')
    model_desc.add_net('fgcnn-ipnn', fg_output.shape, dnn_out.shape)
    return dnn_out


def fibi_nets(embeddings, flatten_emb_layer, dense_layer, concat_emb_dense, config, model_desc):
    """FiBiNet: SENET re-weighted embeddings and the raw embeddings each go through a bilinear interaction;
    the two [B,P,D] blocks are concatenated along the pair axis (deepnets.py:344-371)."""
    senet_index = counter.next_num('senet_layer')
    senet_emb_concat = _concat_embeddings(embeddings, f'concat_senet_embedding_{senet_index}')
    if senet_emb_concat is None:
        model_desc.add_net('fibi', (None), (None))
        return None
    senet_pooling_op = config.fibinet_params.get('senet_pooling_op', 'mean')
    senet_reduction_ratio = config.fibinet_params.get('senet_reduction_ratio', 3)
    bilinear_type = config.fibinet_params.get('bilinear_type', 'field_interaction')
    senet_embedding = layers.SENET(pooling_op=senet_pooling_op, reduction_ratio=senet_reduction_ratio,
                                   name=f'senet_layer_{senet_index}')(senet_emb_concat)
    senet_bilinear_out = layers.BilinearInteraction(bilinear_type=bilinear_type,
                                                    name=f'senet_bilinear_layer_{senet_index}')(senet_embedding)
    bilinear_out = layers.BilinearInteraction(bilinear_type=bilinear_type,
                                              name=f'embedding_bilinear_layer_{senet_index}')(senet_emb_concat)
    concat_bilinear = Concatenate(axis=1, name=f'concat_bilinear_{senet_index}')([senet_bilinear_out, bilinear_out])
    model_desc.add_net('fibi', senet_emb_concat.shape, concat_bilinear.shape)
    return concat_bilinear


def fibi_dnn_nets(embeddings, flatten_emb_layer, dense_layer, concat_emb_dense, config, model_desc):
    """FiBiNet with DNN as deep classifier (deepnets.py:374-386; like the reference it needs a dense input)."""
    if embeddings is None or len(embeddings) <= 1:
        return None
    fibi_output = fibi_nets(embeddings, flatten_emb_layer, dense_layer, concat_emb_dense, config, model_desc)
    dnn_input = Concatenate(name='concat_bilinear_dense')(
        [Flatten(name='flatten_fibi_output')(fibi_output), dense_layer])
    dnn_out = dnn(dnn_input, config.dnn_params, cellname='fibi_dnn')
    model_desc.add_net('fibi-dnn', fibi_output.shape, dnn_out.shape)
    return dnn_out


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
