# -*- coding:utf-8 -*-
"""Drop-in for `deeptables.models.deepmodel.DeepModel` (deeptables/models/deepmodel.py:26-457):
builds inputs -> MultiColumnEmbedding -> BN(concat[flatten_emb, dense]) -> nets -> stacking ->
output with the SAME layer names and graph shape, then drives the train step on MI355X."""
import collections
import math
import os
import pickle
from collections import OrderedDict
from typing import Union

import numpy as np
import torch

from . import deepnets
from .layers import MultiColumnEmbedding, VarLenColumnEmbedding, dt_custom_objects
from .. import functional as F
from .. import training
from ..functional import Dense, Concatenate, Flatten, Input, Add, BatchNormalization, Dropout, Model
from ..utils import consts


def default_device():
    if not torch.cuda.is_available():
        raise RuntimeError('deeptables_amd needs a ROCm GPU (MI355X): the layers hot path has no CPU fallback.')
    return torch.device('cuda', torch.cuda.current_device())


class DeepModel:
    """Class for neural network models (deepmodel.py:26)."""

    def __init__(self, task, num_classes, config, categorical_columns, continuous_columns, model_file=None,
                 var_categorical_len_columns=None, custom_objects=None):
        self.model_desc = ModelDesc()
        self.categorical_columns = categorical_columns
        self.continuous_columns = continuous_columns
        self.var_len_categorical_columns = var_categorical_len_columns
        self.task = task
        self.num_classes = num_classes
        self.config = config
        self.model_file = model_file
        self.model = None
        self.optimizer = None
        self.device = None
        if model_file is not None:
            objs = dict(dt_custom_objects)
            if custom_objects is not None:
                objs.update(custom_objects)
            self.model = self._load_model(model_file, objs)

    # ------------------------------------------------------------------------------------------
    # graph construction (deepmodel.py:259-317)
    # ------------------------------------------------------------------------------------------
    def build(self, device=None):
        self.device = device or default_device()
        self.compiled_loop = None            # a captured loop holds pointers into the model / optimizer it was built on
        if hasattr(self, '_fused_plan'):
            del self._fused_plan
        self.model = self._build_model(self.task, self.num_classes, self.config.nets, self.categorical_columns,
                                       self.continuous_columns, self.config,
                                       self.var_len_categorical_columns).to(self.device)
        self._compile_model(self.model, self.task, self.num_classes, self.config.optimizer, self.config.loss)
        # optional (DT_AMD_FLAT_PARAMS=1): all dense weights live in one flat buffer, their gradients in another (one
        # optimizer launch, one all-reduce; backward kernels accumulate in place); embedding tables stay on their own.
        # Off by default: it removes ~20 launches per step from the layer-by-layer path, yet the DCN step measured
        # 4-10 % SLOWER with it (570-609 vs 548 us graph replay; autograd's accumulate-into-existing-grad path).
        tables = [p for l in self.model.modules() if isinstance(l, (MultiColumnEmbedding, VarLenColumnEmbedding))
                  for p in l.parameters()]
        flat = training.flatten_dense_parameters(self.model, self.optimizer, exclude=tables) \
            if os.environ.get('DT_AMD_FLAT_PARAMS', '0') == '1' else None
        self._generic_flat_grad = None if flat is None else flat[1]
        return self.model

    def _build_model(self, task, num_classes, nets, categorical_columns, continuous_columns, config,
                     var_len_categorical_columns=None):
        F.reset_uids()
        self.model_desc = ModelDesc()
        categorical_inputs, continuous_inputs, var_len_categorical_inputs = \
            self._build_inputs(categorical_columns, continuous_columns, var_len_categorical_columns)
        embeddings = self._build_embeddings(categorical_columns, categorical_inputs, var_len_categorical_columns,
                                            var_len_categorical_inputs, config.embedding_dropout)
        dense_layer = self._build_denses(continuous_columns, continuous_inputs, config.dense_dropout)

        flatten_emb_layer = None
        if len(embeddings) > 0:
            if len(embeddings) == 1:
                flatten_emb_layer = Flatten(name='flatten_embeddings')(embeddings[0])
            else:
                flatten_emb_layer = Flatten(name='flatten_embeddings')(
                    Concatenate(name='concat_embeddings_axis_0', axis=-1)(embeddings))

        self.model_desc.nets = nets
        self.model_desc.stacking = config.stacking_op
        concat_emb_dense = self._concat_emb_dense(flatten_emb_layer, dense_layer)
        outs = OrderedDict()
        for net in nets:
            logit = deepnets.get(net)
            out = logit(embeddings, flatten_emb_layer, dense_layer, concat_emb_dense, self.config, self.model_desc)
            if out is not None:
                outs[net] = out
        if len(outs) > 1:
            logits = []
            for name, out in outs.items():
                if len(out.shape) > 2:
                    out = Flatten(name=f'flatten_{name}_out')(out)
                if out.shape[-1] > 1:
                    logit = Dense(1, use_bias=False, activation=None, name=f'dense_logit_{name}')(out)
                else:
                    logit = out
                logits.append(logit)
            if config.stacking_op == consts.STACKING_OP_ADD:
                x = Add(name='add_logits')(logits)
            elif config.stacking_op == consts.STACKING_OP_CONCAT:
                x = Concatenate(name='concat_logits')(logits)
            else:
                raise ValueError(f'Unsupported stacking_op:{config.stacking_op}.')
        elif len(outs) == 1:
            name, out = outs.popitem()
            if len(out.shape) > 2:
                out = Flatten(name=f'flatten_{name}_out')(out)
            x = out
        else:
            raise ValueError(f'Unexpected logit output.{outs}')
        all_inputs = list(categorical_inputs.values()) + list(var_len_categorical_inputs.values()) + \
            list(continuous_inputs.values())                          # deepmodel.py:310
        output = self._output_layer(x, task, num_classes, use_bias=self.config.output_use_bias)
        return Model(inputs=all_inputs, outputs=output)

    def _compile_model(self, model, task, num_classes, optimizer, loss):
        if loss == 'auto':
            if task in (consts.TASK_BINARY, consts.TASK_MULTILABEL):
                loss_name = 'binary_crossentropy'
            elif task == consts.TASK_REGRESSION:
                loss_name = 'mse'
            elif task == consts.TASK_MULTICLASS:
                loss_name = 'binary_crossentropy' if num_classes == 2 else 'categorical_crossentropy'
            else:
                raise RuntimeError(f'unseen task "{task}"')
        elif isinstance(loss, str):
            loss_name = loss
        else:
            loss_name = getattr(loss, '__name__', 'custom')
            self._custom_loss = loss
        self.loss_name = loss_name
        emb_layers = [l for l in model.layers if isinstance(l, MultiColumnEmbedding)]
        self.optimizer = training.make_optimizer(optimizer, list(model.parameters()), emb_layers)
        self.model_desc.optimizer = self.optimizer
        self.model_desc.loss = loss_name

    def _concat_emb_dense(self, flatten_emb_layer, dense_layer):
        if flatten_emb_layer is not None and dense_layer is not None:
            x = Concatenate(name='concat_embedding_dense')([flatten_emb_layer, dense_layer])
        elif flatten_emb_layer is not None:
            x = flatten_emb_layer
        elif dense_layer is not None:
            x = dense_layer
        else:
            raise ValueError('No input layer exists.')
        x = BatchNormalization(name='bn_concat_emb_dense')(x)
        self.model_desc.set_concat_embed_dense(x.shape)
        return x

    def _build_inputs(self, categorical_columns, continuous_columns, var_len_categorical_columns=None):
        categorical_inputs = OrderedDict()
        continuous_inputs = OrderedDict()
        var_len_categorical_inputs = OrderedDict()
        if categorical_columns is not None and len(categorical_columns) > 0:
            categorical_inputs['all_categorical_vars'] = Input(shape=(len(categorical_columns),),
                                                               name='input_categorical_vars_all')
            self.model_desc.add_input('all_categorical_vars', len(categorical_columns))
        for column in continuous_columns or []:
            continuous_inputs[column.name] = Input(shape=(column.input_dim,), name=column.name, dtype=column.dtype)
            self.model_desc.add_input(column.name, column.input_dim)
        for col in var_len_categorical_columns or []:                 # deepmodel.py:375-378
            var_len_categorical_inputs[col.name] = Input(shape=(col.max_elements_length,), name=col.name)
            self.model_desc.add_input(col.name, col.max_elements_length)
        return categorical_inputs, continuous_inputs, var_len_categorical_inputs

    def _build_embeddings(self, categorical_columns, categorical_inputs, var_len_categorical_columns=None,
                          var_len_inputs=None, embedding_dropout=0.):
        if 'all_categorical_vars' in categorical_inputs:
            input_layer = categorical_inputs['all_categorical_vars']
            input_dims = [column.vocabulary_size for column in categorical_columns]
            output_dims = [column.embeddings_output_dim for column in categorical_columns]
            embeddings = MultiColumnEmbedding(input_dims, output_dims, embedding_dropout,
                                              name=consts.LAYER_PREFIX_EMBEDDING + 'categorical_vars_all',
                                              embeddings_initializer=self.config.embeddings_initializer)(input_layer)
            self.model_desc.set_embeddings(input_dims, output_dims, embedding_dropout)
            embeddings = list(embeddings)
        else:
            embeddings = []
        for column in var_len_categorical_columns or []:              # deepmodel.py:406-418
            embeddings.append(VarLenColumnEmbedding(
                emb_vocab_size=column.vocabulary_size, emb_output_dim=column.embeddings_output_dim,
                dropout_rate=embedding_dropout, name=consts.LAYER_PREFIX_EMBEDDING + column.name,
                embeddings_initializer=self.config.embeddings_initializer, embeddings_regularizer=None,
                activity_regularizer=None)(var_len_inputs[column.name]))
        return embeddings

    def _build_denses(self, continuous_columns, continuous_inputs, dense_dropout, use_batchnormalization=False):
        dense_layer = None
        if continuous_inputs:
            if len(continuous_inputs) > 1:
                dense_layer = Concatenate(name=consts.LAYER_NAME_CONCAT_CONT_INPUTS)(list(continuous_inputs.values()))
            else:
                dense_layer = list(continuous_inputs.values())[0]
        if dense_dropout > 0 and dense_layer is not None:
            dense_layer = Dropout(dense_dropout, name='dropout_dense_input')(dense_layer)
        if use_batchnormalization and dense_layer is not None:
            dense_layer = BatchNormalization(name=consts.LAYER_NAME_BN_DENSE_ALL)(dense_layer)
        self.model_desc.set_dense(dense_dropout, use_batchnormalization)
        return dense_layer

    def _output_layer(self, x, task, num_classes, use_bias=True):
        """Dense(output_dim, name='task_output').  The activation (sigmoid/softmax) is applied by
        `predict`; training evaluates the loss from the logits like Keras does in graph mode."""
        if task == consts.TASK_BINARY:
            activation, output_dim = 'sigmoid', 1
        elif task == consts.TASK_REGRESSION:
            activation, output_dim = None, 1
        elif task == consts.TASK_MULTICLASS:
            if num_classes:
                activation, output_dim = 'softmax', num_classes
            else:
                raise ValueError('"config.multiclass_classes" value must be provided for multi-class task.')
        elif task == consts.TASK_MULTILABEL:
            activation, output_dim = 'sigmoid', num_classes
        else:
            raise ValueError(f'Unknown task type:{task}')
        self.output_activation = activation
        output = Dense(output_dim, activation=None, name='task_output', use_bias=use_bias)(x)
        self.model_desc.set_output(activation, output.shape, use_bias)
        return output

    # ------------------------------------------------------------------------------------------
    # train / predict
    # ------------------------------------------------------------------------------------------
    def _loss(self, logit, y):
        if self.loss_name == 'binary_crossentropy':
            return training.bce_from_logits(logit, y)
        if self.loss_name == 'categorical_crossentropy':
            return training.categorical_ce_from_logits(logit, y)
        if self.loss_name in ('mse', 'mean_squared_error'):
            return training.mse(logit, y)
        if hasattr(self, '_custom_loss'):
            return self._custom_loss(y, self._activate(logit))
        raise ValueError(f'Unsupported loss: {self.loss_name}')

    def _activate(self, logit):
        if self.output_activation == 'sigmoid':
            return torch.sigmoid(logit)
        if self.output_activation == 'softmax':
            return torch.softmax(logit, dim=-1)
        return logit

    def fused_plan(self):
        """Whole-step kernel plan for this graph, if the library has one (deeptables_amd/fused.py)."""
        if not hasattr(self, '_fused_plan'):
            from ..fused import make_fused_plan
            self._fused_plan = make_fused_plan(self)
        return self._fused_plan

    def forward_backward(self, inputs, y, sample_weight=None, logit_out=None):
        """forward -> loss -> backward; gradients land in `.grad` / MultiColumnEmbedding.sparse_grads and stay there until
        the caller's `optimizer.step()`.
        sample_weight [B] (Keras fit's sample_weight x class_weight): the DeepFM / DCN plans scale each row's loss inside
        their loss block; other plans leave a weighted step to the layer-by-layer path.
        logit_out: caller-owned buffer for a fused plan's logits."""
        return self._forward_backward(inputs, y, sample_weight, logit_out=logit_out)

    def _forward_backward(self, inputs, y, sample_weight=None, apply_rows=False, logit_out=None, slot=0, preelected=False,
                          next_ids=None, prepared=False):
        """`forward_backward` with the library-internal switches of `train_step` / compiled.CompiledTrainLoop:
        apply_rows=True is the promise that `self.optimizer.step()` follows immediately: a fused plan may then apply the
        row-sparse update of the table rows looked up once (and the dense update) inside its own kernels — the gradients are
        consumed by the call, so this is not a public mode.  slot / preelected / next_ids / prepared: the compiled loop's
        per-step id buffers and its chained steps (fused.FusedDeepFM.run)."""
        plan = self.fused_plan() if self.model.training else None
        if plan is not None and sample_weight is not None and not getattr(plan, 'takes_sample_weight', False):
            plan = None
        self.optimizer.zero_grad(flat=plan is None) if hasattr(self.optimizer, 'register_flat_group') \
            else self.optimizer.zero_grad()
        self._step_used_plan = plan is not None
        if plan is not None:
            cat = inputs[0]
            dense = inputs[1] if len(inputs) > 1 else None
            kw = {} if sample_weight is None else {'sample_weight': sample_weight}
            if logit_out is not None:
                kw['logit_out'] = logit_out
            if slot or preelected:
                kw['slot'], kw['preelected'] = slot, preelected
            if next_ids is not None or prepared:
                kw['next_ids'], kw['prepared'], kw['slot'] = next_ids, prepared, slot
            loss, logit = plan.run(cat, dense, y, apply_rows=apply_rows, **kw)
            self.model._dt_flat_grad = plan.accum
            return loss[0], logit
        # generic path: the dense gradients accumulate in the model-wide flat buffer (None without one)
        self.model._dt_flat_grad = getattr(self, '_generic_flat_grad', None)
        self.model._dt_sharded_step = False
        logit = self.model(inputs)
        loss = self._loss(logit, y) if sample_weight is None else \
            training.weighted_loss(self.loss_name, logit, y, sample_weight)
        if loss.dim() == 0 and loss.dtype == torch.float32:
            loss.backward(training.unit_grad(loss.device))      # no ones-fill launch (and the BCE skips the multiply)
        else:
            loss.backward()
        return loss.detach(), logit.detach()

    def train_step(self, inputs, y, sample_weight=None):
        """forward -> loss -> backward -> (data-parallel gradient exchange) -> optimizer step."""
        strategy = self.config.distribute_strategy
        loss, logit = self._forward_backward(inputs, y, sample_weight, apply_rows=strategy is None)
        if strategy is not None:
            strategy.exchange_gradients(self.model, self.optimizer)
        self.optimizer.step()
        # a fused plan hands out its static logit buffer (overwritten by the next step): copy it.  Asked of THIS step, not
        # of fused_plan(): a weighted step must not build the plan (it re-homes the tower parameters) as a side effect.
        if getattr(self, '_step_used_plan', False):
            return loss.detach().clone(), logit.detach().clone()    # (the loss is one word of the plan's gradient buffer)
        return loss.detach(), logit.detach()

    def fit(self, X=None, y=None, batch_size=128, epochs=1, verbose=1, callbacks=None, validation_split=0.2,
            validation_data=None, shuffle=True, class_weight=None, sample_weight=None, initial_epoch=0,
            steps_per_epoch=None, validation_steps=None, validation_freq=1, max_queue_size=10, workers=1,
            use_multiprocessing=False, steps_per_execution=None):
        """deepmodel.py:114-129 (`model.fit`).  steps_per_execution (Keras' `model.compile` argument, deepmodel.py:319-346):
        k > 1 trains through compiled.CompiledTrainLoop — k consecutive train steps captured into ONE hipGraph over static
        input slots the device-resident feed fills, replayed steps_per_epoch // k times per epoch; None: the module default
        (`training.DEFAULT_STEPS_PER_EXECUTION`, 'auto' = compiled when the graph has a fused whole-step plan); 1: eager."""
        feed = X if isinstance(X, training.TableBatches) else None    # a ready (device-resident) feed: trained on as it is
        if feed is not None:
            if validation_data is None:
                validation_split = 0
            if class_weight is not None or sample_weight is not None:
                raise ValueError('fit(feed): per-row weights belong into the feed (TableBatches(sample_weight=...))')
        n_fit_rows, tr_i = (feed.n if feed is not None else len(X)), None
        if validation_data is None:
            n = n_fit_rows
            n_val = int(math.ceil(n * validation_split)) if validation_split else 0
            if n_val > 0:
                # under a distribute strategy every rank must hold out the SAME rows (one partition, then sharded)
                st_ = self.config.distribute_strategy
                perm = st_.shared_permutation(n) if (st_ is not None and hasattr(st_, 'shared_permutation')) \
                    else np.random.permutation(n)
                val_i, tr_i = perm[:n_val], perm[n_val:]
                take = (lambda a, i: a.iloc[i]) if hasattr(X, 'iloc') else (lambda a, i: a[i])
                X_val, y_val = take(X, val_i), np.asarray(y)[val_i]
                X, y = take(X, tr_i), np.asarray(y)[tr_i]
            else:
                X_val, y_val = None, None
        else:
            if len(validation_data) != 2:
                raise ValueError(f'Unexpected validation_data length, expected 2 but {len(validation_data)}.')
            X_val, y_val = validation_data[0], validation_data[1]
        if batch_size is None:
            batch_size = 128
        # Keras fit: class_weight {label: w} maps to a per-row weight and multiplies sample_weight (binary / multiclass
        # labels; reference deeptable.py:354-365 passes both through to keras)
        weights = None
        if class_weight is not None or sample_weight is not None:
            weights = np.ones(n_fit_rows if tr_i is None else len(tr_i), dtype=np.float32) if sample_weight is None else \
                np.asarray(sample_weight, dtype=np.float32).reshape(-1).copy()
            if sample_weight is not None:
                # the caller's weights belong to the rows as passed in: checked against THAT length, then sliced with the
                # rows kept for training (a weight vector that merely happens to have the post-split length is an error)
                if len(weights) != n_fit_rows:
                    raise ValueError(f'sample_weight has {len(weights)} entries for {n_fit_rows} rows')
                if tr_i is not None:
                    weights = weights[tr_i]
            if class_weight is not None:
                yl = np.asarray(y)
                yl = yl.argmax(-1) if yl.ndim > 1 and yl.shape[-1] > 1 else yl.reshape(-1)
                cw = {int(k): float(v) for k, v in dict(class_weight).items()}
                weights = weights * np.array([cw.get(int(v), 1.0) for v in yl], dtype=np.float32)
        strategy = self.config.distribute_strategy
        if strategy is not None and not hasattr(strategy, 'exchange_gradients'):
            raise ValueError('[distribute_strategy] in ModelConfig must be a deeptables_amd.parallel.'
                             'DataParallelStrategy (RCCL) instance')
        if self.model is None:
            self.build(strategy.device if strategy is not None else None)
        if strategy is not None:
            strategy.broadcast_parameters(self.model)
            if feed is not None:
                pass                                # a ready feed holds this rank's rows already
            elif weights is not None:
                (X, y), weights = strategy.shard(X, y), strategy.shard(weights, weights)[0]
            else:
                X, y = strategy.shard(X, y)
        train = feed if feed is not None else \
            training.TableBatches(X, y, self.categorical_columns, self.continuous_columns, self.device,
                                  self.task, self.num_classes,
                                  var_len_categorical_columns=self.var_len_categorical_columns,
                                  resident=getattr(self, 'feed_resident', None), sample_weight=weights)
        val = None
        if X_val is not None and len(X_val) > 0:
            val = training.TableBatches(X_val, y_val, self.categorical_columns, self.continuous_columns,
                                        self.device, self.task, self.num_classes,
                                        var_len_categorical_columns=self.var_len_categorical_columns)
        if steps_per_epoch is None:
            steps_per_epoch = max(train.n // batch_size, 1)
        history = training.History()
        metrics = list(self.config.metrics or [])
        stop = False
        for cb in callbacks or []:
            if hasattr(cb, 'set_model'):
                cb.set_model(self)
            if hasattr(cb, 'on_train_begin'):
                cb.on_train_begin()
        from .. import compiled
        bs = min(batch_size, train.n)
        spe = compiled.resolve_steps_per_execution(self, steps_per_execution if steps_per_execution is not None
                                                   else getattr(self, 'steps_per_execution', None),
                                                   train, bs, steps_per_epoch)
        loop = None
        if spe > 1 and train.n >= bs:
            # the same feed object again: the captured graph is reused — as long as everything it holds pointers into is
            # the object it was captured on (the optimizer's slots, the plan's workspace, the model's parameters)
            loop = getattr(self, 'compiled_loop', None)
            if loop is None or loop.feed is not train or loop.B != bs or loop.k != (1 if loop.dp else spe) or \
                    not loop.owned_by(self):
                loop = compiled.CompiledTrainLoop(self, train, bs, spe)
        self.compiled_loop = loop           # (for callers that look: None = eager steps)
        want_out = bool(metrics)
        for epoch in range(initial_epoch, epochs):
            self.model.train()
            losses, probs, ys = [], [], []
            step = 0
            if loop is not None:
                # the compiled loop: the epoch's order is ONE device permutation (drawn exactly where the eager feed draws
                # its own: same batches for the same seed), walked k steps per replay; a pass over the table ends the order
                col = {'loss': losses, 'logit': probs, 'y': ys, 'want_outputs': want_out}
                while step < steps_per_epoch:
                    loop.set_order(train._permutation(None) if shuffle else None)
                    if loop.graph is None and loop.logits is None:
                        step += loop.capture(warm_steps=min(2, steps_per_epoch - step, loop.steps_left()), collect=col)
                    n = min(steps_per_epoch - step, loop.steps_left())
                    if n <= 0:
                        break
                    step += loop.run(n, collect=col)
                losses = [torch.cat([l.reshape(-1) for l in losses])] if losses else []
                probs = [self._activate(p) for p in probs]
            while loop is None and step < steps_per_epoch:
                progressed = False
                for ins, yb in train.iterate(min(batch_size, train.n), shuffle, drop_remainder=True):
                    wb = None
                    if train.weighted:
                        wb, yb = yb[:, -1].contiguous(), yb[:, :-1].contiguous()
                        if train.y_ndim == 1:               # the metric buffers see y as an unweighted fit feeds it
                            yb = yb.reshape(-1)
                    loss, logit = self.train_step(ins, yb, wb)
                    losses.append(loss)
                    probs.append(self._activate(logit))
                    ys.append(yb)
                    step += 1
                    progressed = True
                    if step >= steps_per_epoch:
                        break
                if not progressed:
                    break
            if strategy is not None and getattr(strategy, 'sparse_bucket_ratio', 1.0) < 1.0 and \
                    hasattr(strategy, 'check_sparse_overflow'):
                strategy.check_sparse_overflow()       # a bucket that dropped entries in ANY step of the epoch: fail loudly
            if batch_size > 8192 and getattr(self, '_fused_plan', None) is not None and hasattr(self._fused_plan, 'check_dedupe'):
                self._fused_plan.check_dedupe()        # batches beyond 8192 rows: an election table that overflowed (fused.py)
            logs = {'loss': float(torch.cat([l.reshape(-1) for l in losses]).mean().item()) if losses else float('nan')}
            if probs:
                yp, yt = torch.cat(probs).cpu().numpy(), torch.cat(ys).cpu().numpy()
                for m in metrics:
                    logs[training.metric_name(m)] = training.compute_metric(m, yt, yp, self.task)
            if val is not None and (epoch + 1) % max(validation_freq, 1) == 0:
                self._sync_sharded_tables()        # table rows owned by other ranks: fetch their current values
                vlogs = self._evaluate_batches(val, batch_size, metrics)
                logs.update({f'val_{k}': v for k, v in vlogs.items()})
            history.add(epoch, logs)
            if verbose:
                print(f'Epoch {epoch + 1}/{epochs} - ' + ' - '.join(f'{k}: {v:.4f}' for k, v in logs.items()))
            for cb in callbacks or []:
                if hasattr(cb, 'on_epoch_end'):
                    cb.on_epoch_end(epoch, logs)
                if getattr(cb, 'stop_training', False):
                    stop = True
            if stop:
                break
        self._sync_sharded_tables()
        for cb in callbacks or []:
            if hasattr(cb, 'on_train_end'):
                cb.on_train_end()
        history.history = IgnoreCaseDict(history.history)
        return history

    def _sync_sharded_tables(self):
        """With parallel.ShardedEmbeddingStrategy each rank's packed table is current only for the fields it owns;
        before the model is evaluated / saved every owner broadcasts its rows."""
        strategy = self.config.distribute_strategy
        if strategy is None or not getattr(strategy, 'sharded_embeddings', False) or strategy.world_size == 1:
            return
        if not getattr(self.model, '_dt_sharded_step', False):
            return
        for layer in self.model.modules():
            if isinstance(layer, MultiColumnEmbedding):
                strategy.sync_tables(layer)

    def _evaluate_batches(self, data, batch_size, metrics):
        self.model.eval()
        losses, weights, probs = [], [], []
        with torch.no_grad():
            for ins, yb in data.iterate(batch_size, False, drop_remainder=False):
                logit = self.model(ins)
                losses.append(self._loss(logit, yb))
                weights.append(yb.shape[0])
                probs.append(self._activate(logit))
        w = torch.tensor(weights, dtype=torch.float32, device=self.device)
        logs = {'loss': float((torch.stack(losses) * w).sum().item() / w.sum().item())}
        yp, yt = torch.cat(probs).cpu().numpy(), data.y.cpu().numpy()
        for m in metrics:
            logs[training.metric_name(m)] = training.compute_metric(m, yt, yp, self.task)
        return logs

    def predict(self, X, batch_size=128, verbose=0):
        return self._predict(self.model, X, batch_size=batch_size, activate=True)

    def _predict(self, model, X, batch_size=128, activate=True):
        data = training.TableBatches(X, None, self.categorical_columns, self.continuous_columns, self.device,
                                     var_len_categorical_columns=self.var_len_categorical_columns)
        model.eval()
        outs = []
        with torch.no_grad():
            for ins, _ in data.iterate(batch_size, False, drop_remainder=False):
                o = model(ins)
                if isinstance(o, (list, tuple)):
                    outs.append([t.cpu().numpy() for t in o])
                else:
                    outs.append((self._activate(o) if activate else o).cpu().numpy())
        if outs and isinstance(outs[0], list):
            return [np.concatenate([o[i] for o in outs]) for i in range(len(outs[0]))]
        return np.concatenate(outs) if outs else np.zeros((0, 1), dtype=np.float32)

    def apply(self, X, output_layers=[], concat_outputs=False, batch_size=128, verbose=0, transformer=None):
        if len(output_layers) <= 0:
            raise ValueError('"output_layers" at least 1 element.')
        outputs = [self.model.get_layer(l).output for l in output_layers]
        if len(outputs) > 1 and concat_outputs:
            outputs = Concatenate()(outputs)
        proxy = Model(inputs=self.model.inputs, outputs=outputs if len(output_layers) > 1 else outputs[0])
        output = self._predict(proxy, X, batch_size=batch_size, activate=False)
        if transformer is None:
            return output
        if isinstance(output, list):
            return [transformer.fit_transform(o.reshape((o.shape[0], -1))) for o in output]
        return transformer.fit_transform(output)

    def evaluate(self, X_test, y_test, batch_size=256, verbose=0, return_dict=True):
        data = training.TableBatches(X_test, y_test, self.categorical_columns, self.continuous_columns, self.device,
                                     self.task, self.num_classes,
                                     var_len_categorical_columns=self.var_len_categorical_columns)
        result = self._evaluate_batches(data, batch_size, list(self.config.metrics or []))
        return IgnoreCaseDict(inputs=result) if return_dict else list(result.values())

    # ------------------------------------------------------------------------------------------
    # persistence: state dict with Keras-style names (h5py is not available; deepmodel.py:205-221)
    # ------------------------------------------------------------------------------------------
    def save(self, filepath, include_optimizer=False):
        """deepmodel.py:205-210 saves the Keras model as .h5; here: one safetensors file with the Keras weight
        names (deeptables_amd/checkpoint.py), optionally with the optimizer slots."""
        from .. import checkpoint
        st = self.config.distribute_strategy
        if include_optimizer and getattr(st, 'sharded_embeddings', False) and getattr(st, 'active', False):
            # row-owned tables: a rank holds Adam moments only for the fields it owns (sync_tables restores the weights,
            # not the slots), so one rank's file would reset every other rank's moments on resume
            raise ValueError('save(include_optimizer=True) under ShardedEmbeddingStrategy: every rank holds the Adam '
                             'moments of its own fields only; save without the optimizer, or train with replicated tables')
        os.makedirs(os.path.dirname(os.path.abspath(filepath)) or '.', exist_ok=True)
        checkpoint.save_model(self.model, filepath, optimizer=self.optimizer if include_optimizer else None,
                              metadata={'task': str(self.task), 'nets': [str(n) for n in self.config.nets]})

    def _load_model(self, filepath, custom_objects=None):
        from .. import checkpoint
        model = self.build()
        if str(filepath).endswith('.h5'):
            checkpoint.import_keras_h5(model, filepath)
        else:
            checkpoint.load_model(model, filepath, optimizer=self.optimizer)
        return model

    def release(self):
        self.model = None
        self.optimizer = None
        self.compiled_loop = None


class ModelDesc:
    """Human-readable description of the assembled graph (deepmodel.py:460-532)."""

    def __init__(self):
        self.inputs = []
        self.embeddings = None
        self.dense = None
        self.concat_embed_dense = None
        self.nets = []
        self.nets_info = []
        self.stacking = None
        self.output = None
        self.loss = None
        self.optimizer = None

    def add_input(self, name, num_columns):
        self.inputs.append(f'{name}: ({num_columns})')

    def set_embeddings(self, input_dims, output_dims, embedding_dropout):
        self.embeddings = f'input_dims: {input_dims}\noutput_dims: {output_dims}\ndropout: {embedding_dropout}'

    def set_dense(self, dense_dropout, use_batchnormalization):
        self.dense = f'dropout: {dense_dropout}\nbatch_normalization: {use_batchnormalization}'

    def set_concat_embed_dense(self, output_shape):
        self.concat_embed_dense = f'shape: {output_shape}'

    def add_net(self, name, input_shape, output_shape):
        self.nets_info.append(f'{name}: input_shape {input_shape}, output_shape {output_shape}')

    def set_output(self, activation, output_shape, use_bias):
        self.output = f'activation: {activation}, output_shape: {output_shape}, use_bias: {use_bias}'

    def nets_desc(self):
        return '\n'.join(self.nets_info)

    def optimizer_info(self):
        if self.optimizer is None:
            return None
        return getattr(self.optimizer, '_name', self.optimizer)

    def __str__(self):
        bar = '-' * 57
        rows = ['>>>>>>>>>>>>>>>>>>>>>> Model Desc <<<<<<<<<<<<<<<<<<<<<<< ', bar, 'inputs:', bar,
                f'{[c for c in self.inputs]}', bar, 'embeddings:', bar, f'{self.embeddings}', bar,
                f'dense: {self.dense}', bar, f'concat_embed_dense: {self.concat_embed_dense}', bar,
                f'nets: {self.nets}', bar, f'{self.nets_desc()}', bar, f'stacking_op: {self.stacking}', bar,
                f'output: {self.output}', f'loss: {self.loss}', f'optimizer: {self.optimizer_info()}', bar, '']
        return '\n'.join(rows)


class IgnoreCaseDict(collections.UserDict):
    """Metric dict with case-insensitive string keys (deepmodel.py:535-563)."""

    def __init__(self, inputs: Union[dict, collections.UserDict] = None):
        super().__init__(inputs.data if isinstance(inputs, collections.UserDict) else inputs)
        for k in self.data:
            if not isinstance(k, str):
                raise KeyError(f"Key should be str but is {k}")
        self.data.update({k.lower(): self.data[k] for k in list(self.data)})

    @staticmethod
    def _key(item):
        if not isinstance(item, str):
            raise KeyError(f"Key should be str but is {item}")
        return item.lower()

    def __contains__(self, item):
        return self._key(item) in self.data

    def __setitem__(self, item, value):
        self.data[self._key(item)] = value

    def __getitem__(self, item):
        return self.data[self._key(item)]
