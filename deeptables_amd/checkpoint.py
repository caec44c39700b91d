# -*- coding:utf-8 -*-
"""Checkpoint format: one safetensors file holding every weight of the model under its
KERAS name — `<layer name>/<weight name>` exactly as `keras.Model.save_weights` of the reference graph would
name them (deepmodel.py:205-221 saves an .h5 of that model; deeptable.py:773-804 wraps it) — plus, optionally,
the optimizer's slots (`optimizer/<weight>/m|v`, step count in the metadata).

The packed embedding table of `MultiColumnEmbedding` is stored as the reference's per-column variables
`embeddings_{i}` (layers.py:863-877); Cross / CIN / BilinearInteraction parameter lists get the reference's
indexed names (`kernels_{i}`, `bias_{i}`, `f_{i}`, `bilinear_weight{i}_{j}`); nested Keras layers (the Dense /
BatchNormalization inside MultiheadAttention, CIN, AFM, SENET, FGCNN) become `<layer>/<sublayer>/<weight>`.
Tensors are written through views of the live parameters, so saving never materialises a second copy of a
multi-GB table on the device (safetensors serialises from host memory; the table crosses PCIe once).

`import_keras_h5` reads a Keras `save_weights` .h5 of the REFERENCE model into the same name space when h5py
is installed (it is not in this image — the function raises ImportError).
"""
import json
import re
from collections import OrderedDict

import torch

FORMAT = 'deeptables_amd.safetensors.v1'

_SUFFIX_RULES = (
    (re.compile(r'^(kernels|bias)\.(\d+)$'), r'\1_\2'),      # Cross: kernels_0 / bias_0 (layers.py:423-426)
    (re.compile(r'^f_\.(\d+)$'), r'f_\1'),                   # CIN filters (layers.py:659)
    (re.compile(r'^f0_\.(\d+)$'), r'f0_\1'),                 # CIN reduce_D factors (layers.py:652-655)
    (re.compile(r'^f__\.(\d+)$'), r'f__\1'),
)


def _keras_weight_name(cls, pname):
    if cls == 'CIN' and pname.startswith('bias.'):
        return 'bias' + pname.split('.')[1]                  # layers.py:662 names them bias{i}
    for rx, rep in _SUFFIX_RULES:
        if rx.match(pname):
            return rx.sub(rep, pname)
    if cls == 'FGCNN':
        pname = {'conv2d_kernel': 'conv2d/kernel', 'conv2d_bias': 'conv2d/bias'}.get(pname, pname)
    return pname.replace('.', '/')                           # nested layer: dense_Q.kernel -> dense_Q/kernel


def named_weights(model):
    """-> OrderedDict {keras name: tensor VIEW of the live parameter / buffer} in graph order."""
    out = OrderedDict()
    for lname, layer in model.layers_by_name.items():
        cls = layer.__class__.__name__
        if cls == 'MultiColumnEmbedding':
            for i, e in enumerate(layer.embeddings):
                out[f'{lname}/embeddings_{i}'] = e
            continue
        if cls == 'BilinearInteraction':
            for k, wname in enumerate(layer.weight_names()):
                out[f'{lname}/{wname}'] = layer.W[k]
            continue
        if cls == 'Cross':
            for i in range(layer.num_cross_layer):
                out[f'{lname}/kernels_{i}'] = layer.kernel_stack[i].view(-1, 1)     # Keras shape (num_dims, 1)
            for i in range(layer.num_cross_layer):
                out[f'{lname}/bias_{i}'] = layer.bias_stack[i].view(-1, 1)
            continue
        if cls == 'VarLenColumnEmbedding':
            out[f'{lname}/embedding/embeddings'] = layer.embeddings
            continue
        for pname, p in list(layer.named_parameters()) + list(layer.named_buffers()):
            out[f'{lname}/{_keras_weight_name(cls, pname)}'] = p
    return out


def _slot_piece(slot, off, t):
    """the part of an optimizer slot that belongs to variable `t`, which starts `off` elements into the parameter the
    slot mirrors — as a VIEW (slots of packed tables can be row-strided, so no reshape(-1) which would copy)"""
    if off == 0 and tuple(slot.shape) == tuple(t.shape):
        return slot
    if slot.dim() == 2 and t.dim() == 2 and off % slot.shape[1] == 0 and t.shape[1] == slot.shape[1]:
        r0 = off // slot.shape[1]
        return slot[r0:r0 + t.shape[0]]
    return slot.reshape(-1)[off:off + t.numel()].reshape(t.shape)


def save_model(model, path, optimizer=None, metadata=None):
    """Write `path` (safetensors).  optimizer: a KerasAdam whose m/v slots are stored next to the weights."""
    from safetensors.torch import save_file
    tensors = OrderedDict()
    for name, t in named_weights(model).items():
        tensors[name] = t.detach().to('cpu').contiguous()
    meta = {'format': FORMAT}
    if optimizer is not None and hasattr(optimizer, 'state'):
        by_id = {id(t): name for name, t in named_weights(model).items()}
        by_ptr = {}
        for name, t in named_weights(model).items():
            by_ptr.setdefault(t.untyped_storage().data_ptr(), []).append((name, t))
        for p in list(getattr(optimizer, 'params', [])):
            st = optimizer.state.get(id(p))
            if st is None:
                continue
            # a parameter may be stored as several Keras variables (packed table -> embeddings_i): slice the slots
            views = [(n, t) for n, t in by_ptr.get(p.untyped_storage().data_ptr(), [])
                     if _inside(t, p)]
            for name, t in views:
                off = (t.data_ptr() - p.data_ptr()) // 4
                for slot in ('m', 'v'):          # (a table's slots may be strided views of one [V,2,D] array)
                    tensors[f'optimizer/{name}/{slot}'] = _slot_piece(st[slot], off, t).detach().to('cpu').contiguous()
        meta['optimizer'] = getattr(optimizer, '_name', optimizer.__class__.__name__)
        meta['optimizer_iterations'] = str(int(getattr(optimizer, 't', 0)))
        del by_id
    if metadata:
        meta.update({k: v if isinstance(v, str) else json.dumps(v) for k, v in metadata.items()})
    save_file(tensors, path, metadata=meta)
    return list(tensors)


def _same_view(a, b):
    return a.data_ptr() == b.data_ptr() and tuple(a.shape) == tuple(b.shape) and a.stride() == b.stride()


def _inside(view, base):
    if _same_view(view, base):          # the variable IS the parameter (possibly a strided block of a fused plan's slab)
        return view.is_floating_point()
    lo, hi = base.data_ptr(), base.data_ptr() + base.numel() * base.element_size()
    return view.is_floating_point() and lo <= view.data_ptr() < hi and view.is_contiguous()


def load_model(model, path, optimizer=None, strict=True):
    """Copy the file's tensors into the live parameters (in place: packed tables, flat buffers and captured graphs
    keep their addresses).  Returns the file's metadata."""
    from safetensors import safe_open
    targets = named_weights(model)
    seen = set()
    with safe_open(path, framework='pt', device='cpu') as f:
        meta = f.metadata() or {}
        if meta.get('format') != FORMAT:
            raise ValueError(f'{path}: not a {FORMAT} file (format={meta.get("format")!r})')
        keys = set(f.keys())
        with torch.no_grad():
            for name, t in targets.items():
                if name not in keys:
                    if strict:
                        raise KeyError(f'{path}: missing weight {name!r}')
                    continue
                src = f.get_tensor(name)
                if tuple(src.shape) != tuple(t.shape):
                    raise ValueError(f'{name}: file has shape {tuple(src.shape)}, model {tuple(t.shape)}')
                t.copy_(src.to(t.dtype))
                seen.add(name)
            if optimizer is not None and meta.get('optimizer') and hasattr(optimizer, 'state'):
                by_ptr = {}
                for name, t in targets.items():
                    by_ptr.setdefault(t.untyped_storage().data_ptr(), []).append((name, t))
                row_sparse = {id(t) for layer in getattr(optimizer, 'embedding_layers', ())
                              for t in layer.tables.values() if not layer.uses_dense_grad(t.shape[1])}
                for p in optimizer.params:
                    # never convert a slot layout while loading: a table that already holds the interleaved [V,2,D] slot
                    # record keeps it (captured graphs hold its address; `_slot_piece` reads through row-strided views),
                    # and state created here for a packed 2-D table is created in the layout the row update uses
                    st = optimizer.state.get(id(p))
                    if st is None:
                        st = optimizer._st(p, rows=id(p) in row_sparse)
                    for name, t in by_ptr.get(p.untyped_storage().data_ptr(), []):
                        if not _inside(t, p) or f'optimizer/{name}/m' not in keys:
                            continue
                        off = (t.data_ptr() - p.data_ptr()) // 4
                        for slot in ('m', 'v'):
                            dst = _slot_piece(st[slot], off, t)
                            dst.copy_(f.get_tensor(f'optimizer/{name}/{slot}').reshape(dst.shape))
                optimizer.t = int(meta.get('optimizer_iterations', '0'))
    extra = [k for k in keys if not k.startswith('optimizer/') and k not in seen]
    if strict and extra:
        raise KeyError(f'{path}: unexpected weights {extra[:5]}{"..." if len(extra) > 5 else ""}')
    return meta


def import_keras_h5(model, path, strict=True):
    """Load a Keras `model.save_weights('x.h5')` / `model.save('x.h5')` of the reference graph (same layer names)
    into `model`.  Needs h5py."""
    try:
        import h5py
    except ImportError as e:      # not installable in this environment
        raise ImportError('import_keras_h5 needs h5py, which is not available here') from e
    import numpy as np
    found = {}
    with h5py.File(path, 'r') as f:
        root = f['model_weights'] if 'model_weights' in f else f

        def visit(name, obj):
            if isinstance(obj, h5py.Dataset):
                parts = [p for p in name.split('/') if p]
                # <layer>/<layer>/<weight>:0 (tf.keras) or <layer>/vars/<i> (keras 3 .weights.h5 is not name based)
                if len(parts) >= 2:
                    layer = parts[0]
                    wname = '/'.join(parts[2:] if len(parts) > 2 and parts[1] == layer else parts[1:])
                    found[f'{layer}/{wname.split(":")[0]}'] = np.asarray(obj)
        root.visititems(visit)
    targets = named_weights(model)
    with torch.no_grad():
        for name, t in targets.items():
            if name not in found:
                if strict and t.is_floating_point() and not name.endswith(('row_offset', 'oob_count')):
                    raise KeyError(f'{path}: missing weight {name!r}')
                continue
            t.copy_(torch.as_tensor(found[name]).reshape(t.shape).to(t.dtype))
    return sorted(found)
