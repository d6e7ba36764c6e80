"""
Functional `Model` of the shim: graph of shared layer instances -> execution plan with the cubed-sphere fusions ->
HIP kernels.  Host API subset used by the reference: `Model(inputs, outputs)`, `.outputs`, `.layers`,
`.compile(loss, loss_weights, optimizer, metrics)`, `.fit`, `.predict`, `.evaluate`, `.summary`, `.save`,
`.save_weights`, `.load_weights`, `.get_weights`, `.set_weights`, `.stop_training`, `.optimizer`.

Execution model (MI355X-first, replaces TF's graph executor):
  * plan: the layer graph is flattened once; `CubeSpherePadding2D -> CubeSphereConv2D [-> ReLU]`, optionally fed by
    `UpSampling3D` and/or `concatenate`, becomes ONE fused kernel launch (halo gather, upsample and concat resolved in
    the convolution's load, bias + ReLU(0.1,10) in its epilogue) -- nothing padded/upsampled/concatenated is materialised;
  * parameters of all layers are views into one flat fp32 buffer, gradients into a second one: a single Adam kernel and a
    single RCCL all-reduce (data parallel, one process per GPU) serve the whole model;
  * a training step with static shapes is captured into a hipGraph (torch.cuda.CUDAGraph) after one eager warm-up step
    and replayed, so the ~150 kernel launches of a step cost one graph launch on the host.
"""
import json
import os
import pickle
import time

import numpy as np
import torch

from .. import ops, parallel
from .._native import ACT_LEAKY_CLIP, ACT_NONE
from . import backend, callbacks as cbks, optimizers
from .engine import KTensor, Layer
from .layers import AveragePooling3D, Concatenate, InputLayer, ReLU, UpSampling3D


def _as_list(x):
    if isinstance(x, (list, tuple)):
        return list(x)
    return [x]


class Model(object):
    def __init__(self, inputs=None, outputs=None, name=None):
        if inputs is None or outputs is None:
            raise ValueError('Model needs `inputs` and `outputs`')
        self._single_input = not isinstance(inputs, (list, tuple))
        self.inputs = _as_list(inputs)
        self.outputs = _as_list(outputs)
        self._single_output = len(self.outputs) == 1       # keras unpacks one-element output lists
        for t in self.inputs + self.outputs:
            if not isinstance(t, KTensor):
                raise ValueError('Model inputs/outputs must be symbolic tensors created by Input() and layer calls')
        self.name = name or 'model'
        self.stop_training = False
        self.optimizer = None
        self.loss = None
        self.loss_weights = None
        self.metrics = []
        self.history = None
        self.use_graphs = os.environ.get('DLWPCS_GRAPHS', '1') != '0'
        # activation dtype on the device ('float32' | 'bfloat16'); parameters / gradients / Adam state are always fp32
        self.compute_dtype = backend.compute_dtype()
        self.prepack_weights = os.environ.get('DLWPCS_PREPACK', '1') != '0'
        self.wgrad_side_stream = os.environ.get('DLWPCS_SIDE_STREAM', '0') == '1'   # measured slower on MI355X: off
        # one reduction launch for all layers' weight-gradient partials (DLWPCS_CONV_DEFER_REDUCE)
        self.defer_wgrad_reduce = os.environ.get('DLWPCS_DEFER_REDUCE', '1') == '1'
        # True: the caller feeds every step through the SAME device tensors (e.g. a generator that assembles each batch in
        # place): the captured graphs read them directly instead of copying each batch into private static buffers
        self.static_batch_buffers = False
        self._ones = {}
        self._compiled = False
        self._flat_params = self._flat_grads = None
        self._grads_clean = False
        self._graphs = {}
        self._seen_batch = {}
        self._toposort()
        self._build_plan()

    # -------------------------------------------------------------------------------------------------------------- #
    # graph
    # -------------------------------------------------------------------------------------------------------------- #
    def _toposort(self):
        order, seen = [], set()

        def visit(t):
            if t.uid in seen:
                return
            seen.add(t.uid)
            for i in t.node_inputs:
                visit(i)
            order.append(t)
        for o in self.outputs:
            visit(o)
        input_uids = {t.uid for t in self.inputs}
        for t in order:
            if isinstance(t.layer, InputLayer) and t.uid not in input_uids:
                raise ValueError('Graph disconnected: cannot obtain value for tensor %r' % (t,))
        self._nodes = order
        layers, seen_l = [], set()
        for t in self.inputs + order:
            if t.layer is not None and id(t.layer) not in seen_l:
                seen_l.add(id(t.layer))
                layers.append(t.layer)
        self.layers = layers

    def _build_plan(self):
        consumers = {}
        for t in self._nodes:
            for i in t.node_inputs:
                consumers.setdefault(i.uid, []).append(t)
        out_uids = {o.uid for o in self.outputs}

        def sole_consumer(t):
            c = consumers.get(t.uid, [])
            return c[0] if (len(c) == 1 and t.uid not in out_uids) else None

        from ..custom import CubeSphereConv2D, CubeSpherePadding2D
        virtual = set()          # tensors never materialised
        fused = {}               # uid of the tensor a fused step produces -> step description
        for t in self._nodes:
            lay = t.layer
            if not isinstance(lay, CubeSphereConv2D) or len(t.node_inputs) != 1:
                continue
            padt = t.node_inputs[0]
            if not isinstance(padt.layer, CubeSpherePadding2D) or sole_consumer(padt) is not t:
                continue
            if padt.layer.data_format != 'channels_last' or not lay.can_fuse_halo(padt.layer.padding[1][0]):
                continue
            if padt.layer.padding[1] != padt.layer.padding[2] or padt.layer.padding[1][0] != padt.layer.padding[1][1]:
                continue
            x = padt.node_inputs[0]
            src0, src1, up0 = x, None, False
            chain = [padt]
            if isinstance(x.layer, Concatenate) and len(x.node_inputs) == 2 and sole_consumer(x) is padt \
                    and x.layer._axis(len(x.shape)) == len(x.shape) - 1:
                src0, src1 = x.node_inputs
                chain.append(x)
            if isinstance(src0.layer, UpSampling3D) and sole_consumer(src0) is (chain[-1]) \
                    and src0.layer.size == (1, 2, 2):
                chain.append(src0)
                src0 = src0.node_inputs[0]
                up0 = True
            out_t, act, alpha, vmax = t, ACT_NONE, 0.0, 0.0
            nxt = sole_consumer(t)
            if nxt is not None and isinstance(nxt.layer, ReLU) and nxt.layer.threshold == 0.:
                out_t, act = nxt, ACT_LEAKY_CLIP
                alpha = nxt.layer.negative_slope
                vmax = float('inf') if nxt.layer.max_value is None else nxt.layer.max_value
                virtual.add(t.uid)
            for c in chain:
                virtual.add(c.uid)
            fused[out_t.uid] = ('fused_conv', out_t.uid, lay, src0.uid, None if src1 is None else src1.uid, up0, act,
                                alpha, vmax)
        steps = []
        for t in self._nodes:
            if isinstance(t.layer, InputLayer) or t.uid in virtual and t.uid not in fused:
                continue
            if t.uid in fused:
                steps.append(fused[t.uid])
            elif isinstance(t.layer, AveragePooling3D) and len(t.node_inputs) == 1 and (
                    len(consumers.get(t.node_inputs[0].uid, [])) > 1 or t.node_inputs[0].uid in out_uids):
                # the pooled tensor has other consumers (U-Net skip connection): they are re-routed through an alias so
                # that both gradients reach ONE backward kernel (ops.avgpool2_skip)
                steps.append(('pool_skip', t.uid, t.layer, t.node_inputs[0].uid))
            else:
                steps.append(('layer', t.uid, t.layer, [i.uid for i in t.node_inputs],
                              isinstance(t.layer, Concatenate)))
        self._plan = steps
        self.n_fused = len(fused)

    def _pack_state(self, device):
        """Packed-weight buffers + the device item table of every matrix-core convolution layer (built once per
        (device, compute dtype); the parameter tensors never move after _flatten_parameters)."""
        from ..custom import CubeSphereConv2D
        from .. import _native as nat
        tag = nat.BF16 if self.compute_dtype == 'bfloat16' else nat.F32
        key = (str(device), tag, self._flat_params.data_ptr() if self._flat_params is not None else 0)
        st = getattr(self, '_pack_cache', None)
        if st is not None and st['key'] == key:
            return st
        entries, table = [], {}
        for lay in self.layers:
            if not isinstance(lay, CubeSphereConv2D) or not lay.built or not lay._is_mfma_config():
                continue
            we = lay.equatorial_kernel
            if we.device != device or id(we) in table:
                continue
            bufs = ops.conv_packed_buffers(lay.kernel_size[0], we.shape[2], we.shape[3], tag, device,
                                           bias=lay.equatorial_bias is not None)
            entries.append((we, lay.polar_kernel, lay.north_pole_kernel, lay.equatorial_bias, lay.polar_bias,
                            lay.north_pole_bias, bufs, lay.kernel_size[0], lay.flip_north_pole, tag))
            table[id(we)] = (tag, bufs[0], bufs[1], bufs[2])
        st = {'key': key, 'items': ops.make_pack_items(entries, device) if entries else None, 'n': len(entries),
              'table': table, 'keep': entries}
        self._pack_cache = st
        return st

    def _forward(self, inputs, repack=True):
        want = backend.torch_dtype(self.compute_dtype)
        inputs = [v if v.dtype == want else v.to(want) for v in inputs]
        if inputs and inputs[0].is_cuda and self.prepack_weights:
            # one launch packs the weights of every layer for this pass (they changed with the last optimizer step);
            # repack=False: the caller knows they have not changed since its previous pass (steps of one rollout)
            st = self._pack_state(inputs[0].device)
            if st['n'] and (repack or not st.get('packed')):
                ops.pack_batch(st['items'], st['n'])
                st['packed'] = True
            ops.PREPACKED = st['table']
            try:
                return self._run_plan(inputs)
            finally:
                ops.PREPACKED = {}
        return self._run_plan(inputs)

    def _run_plan(self, inputs):
        values = {t.uid: v for t, v in zip(self.inputs, inputs)}
        for st in self._plan:
            if st[0] == 'fused_conv':
                _, out_uid, lay, s0, s1, up0, act, alpha, vmax = st
                values[out_uid] = lay.fused_call(values[s0], None if s1 is None else values[s1], up0=up0, halo=True,
                                                 act=act, alpha=alpha, vmax=vmax)
            elif st[0] == 'pool_skip':
                _, out_uid, lay, in_uid = st
                values[out_uid], values[in_uid] = ops.avgpool2_skip(values[in_uid])
            else:
                _, out_uid, lay, in_uids, takes_list = st
                args = [values[u] for u in in_uids]
                values[out_uid] = lay.call(args if (takes_list or len(args) > 1) else args[0])
        return [values[o.uid] for o in self.outputs]

    def __call__(self, inputs):
        """Eager application on device tensors (differentiable)."""
        self._grads_clean = False           # a backward through this call accumulates into the flat gradient buffer
        outs = self._forward(_as_list(inputs))
        return outs[0] if self._single_output else outs

    # -------------------------------------------------------------------------------------------------------------- #
    # weights
    # -------------------------------------------------------------------------------------------------------------- #
    def _weight_layers(self):
        return [l for l in self.layers if l._weights]

    @property
    def weights(self):
        return [w for l in self._weight_layers() for w in l._weights]

    @property
    def trainable_weights(self):
        return [w for w in self.weights if w.requires_grad]

    def count_params(self):
        return int(sum(w.numel() for w in self.weights))

    def get_weights(self):
        return [a for l in self._weight_layers() for a in l.get_weights()]

    def set_weights(self, weights):
        weights = list(weights)
        n = sum(len(l._weights) for l in self._weight_layers())
        if len(weights) != n:
            raise ValueError('You called `set_weights(weights)` on model "%s" with a weight list of length %d, but the '
                             'model was expecting %d weights.' % (self.name, len(weights), n))
        k = 0
        for l in self._weight_layers():
            l.set_weights(weights[k:k + len(l._weights)])
            k += len(l._weights)

    def _flatten_parameters(self):
        """Move every weight into one flat fp32 buffer (+ a flat gradient buffer); layers keep views."""
        ws = self.weights
        dev = backend.device()
        total = sum(w.numel() for w in ws)
        # 64-element (256 B) alignment of every weight keeps float4 loads in the kernels aligned
        offsets, off = [], 0
        for w in ws:
            offsets.append(off)
            off += (w.numel() + 63) // 64 * 64
        flat = torch.zeros(off, dtype=torch.float32, device=dev)
        grads = torch.zeros(off, dtype=torch.float32, device=dev)
        k = 0
        for l in self._weight_layers():
            new = []
            for w in l._weights:
                o = offsets[k]
                view = flat[o:o + w.numel()].view(w.shape)
                with torch.no_grad():
                    view.copy_(w.detach().to(dev))
                view.requires_grad_(w.requires_grad)
                if w.requires_grad:
                    view.grad = grads[o:o + w.numel()].view(w.shape)
                new.append(view)
                k += 1
            l._rebind(new)
        self._flat_params, self._flat_grads = flat, grads
        self._n_params = total

    # -------------------------------------------------------------------------------------------------------------- #
    # compile / train
    # -------------------------------------------------------------------------------------------------------------- #
    def compile(self, optimizer='adam', loss=None, metrics=None, loss_weights=None, **kwargs):
        self.optimizer = optimizers.get(optimizer)
        mp_dtype = getattr(self.optimizer, '_mixed_precision', None)
        if mp_dtype is not None and mp_dtype != self.compute_dtype:
            # optimizer came out of enable_mixed_precision_graph_rewrite() AFTER this model was constructed
            self.compute_dtype = mp_dtype
            self._pack_cache = None
        n_out = len(self.outputs)
        losses = _as_list(loss) if isinstance(loss, (list, tuple)) else [loss] * n_out
        for l in losses:
            if l not in ('mse', 'mean_squared_error', 'MSE'):
                raise NotImplementedError("loss %r: the DLWP-CS engine provides 'mse' (reference Azure/train_cs.py:424)"
                                          % (l,))
        self.loss = loss
        if loss_weights is None:
            self.loss_weights = [1.0] * n_out
        else:
            self.loss_weights = [float(v) for v in _as_list(loss_weights)]
            if len(self.loss_weights) != n_out:
                raise ValueError('When passing a list as loss_weights, it should have one entry per model output. The '
                                 'model has %d outputs, but you passed loss_weights=%s' % (n_out, loss_weights))
        metrics = _as_list(metrics) if metrics else []
        for m in metrics:
            if m not in ('mae', 'mean_absolute_error'):
                raise NotImplementedError("metric %r: the DLWP-CS engine provides 'mae'" % (m,))
        self.metrics = metrics
        self._flatten_parameters()
        self._graphs.clear()
        self._seen_batch.clear()
        self._world = parallel.world()[1]
        parallel.broadcast_parameters(self._flat_params)     # identical replicas: rank 0's initial weights everywhere
        self._compiled = True

    def _metric_names(self):
        """keras log names: 'loss', per-output losses (multi-output only), then metrics."""
        names = ['loss']
        outs = [o.layer.name for o in self.outputs]
        if len(self.outputs) > 1:
            # keras uniquifies repeated output-layer names
            seen, uniq = {}, []
            for n in outs:
                k = seen.get(n, 0)
                seen[n] = k + 1
                uniq.append(n if k == 0 else '%s_%d' % (n, k))
            names += ['%s_loss' % n for n in uniq]
            if self.metrics:
                names += ['%s_mean_absolute_error' % n for n in uniq]
        elif self.metrics:
            names += ['mean_absolute_error']
        return names

    def _assemble_logs(self, sums, count):
        """sums: (n_out, 2) tensor of [weighted mse, mae] sums over `count` batches -> ordered keras values."""
        s = (sums / max(count, 1)).cpu().numpy()
        w = np.asarray(self.loss_weights, dtype=np.float64)
        vals = [float(s[:, 0].sum())]
        if len(self.outputs) > 1:
            vals += [float(s[i, 0] / w[i]) if w[i] != 0 else 0.0 for i in range(len(self.outputs))]
            if self.metrics:
                vals += [float(s[i, 1]) for i in range(len(self.outputs))]
        elif self.metrics:
            vals += [float(s[0, 1])]
        return vals

    def _loss_and_backward(self, inputs, targets, train=True):
        outs = self._forward(inputs)
        if len(targets) != len(outs):
            raise ValueError('Error when checking model target: expected %d target arrays, got %d'
                             % (len(outs), len(targets)))
        stats = [ops.mse_mae(o, t, w) for o, t, w in zip(outs, targets, self.loss_weights)]
        if train:
            dev = stats[0].device
            ones = [ops.unit_seed(dev) for _ in stats]
            ops.DIRECT_PARAM_GRADS = True       # weight gradients accumulate straight into the flat gradient buffer
            ops.WGRAD_SIDE_STREAM = self.wgrad_side_stream
            ops.DEFER_WGRAD_REDUCE = self.defer_wgrad_reduce    # ... through ONE reduction launch for all layers
            ops.drop_deferred_reduce()
            try:
                torch.autograd.backward(stats, ones)
                ops.flush_deferred_reduce(dev)
            finally:
                ops.DIRECT_PARAM_GRADS = False
                ops.WGRAD_SIDE_STREAM = False
                ops.DEFER_WGRAD_REDUCE = False
                ops.drop_deferred_reduce()
                ops.join_side_stream(stats[0].device)
        if len(stats) == 1:
            return stats[0].detach().view(1, 2)                 # no copy launch for the single-output case
        return torch.stack([s.detach() for s in stats])

    def _apply_gradients(self):
        scale = parallel.allreduce_gradients(self._flat_grads)  # RCCL over xGMI: one flat 2.7 MB buffer per step
        self.optimizer.apply(self._flat_params, self._flat_grads, grad_scale=scale)

    def _train_step_eager(self, inputs, targets):
        self._flat_grads.zero_()
        self._grads_clean = False                               # the gradients stay readable after an eager step
        stats = self._loss_and_backward(inputs, targets, True)
        self._apply_gradients()
        return stats

    def train_on_device_batch(self, inputs, targets):
        """
        One optimisation step on device-resident tensors.  Static shapes are captured in a hipGraph on their second
        occurrence and replayed afterwards.  Returns the (n_out, 2) device tensor of [weighted mse, mae].
        """
        if not self._compiled:
            raise RuntimeError('You must compile your model before training/testing. Use `model.compile(...)`.')
        key = tuple(tuple(t.shape) for t in inputs + targets)
        if not self.use_graphs:
            return self._train_step_eager(inputs, targets)
        g = self._graphs.get(key)
        if g is None:
            n = self._seen_batch.get(key, 0)
            self._seen_batch[key] = n + 1
            if n == 0:
                return self._train_step_eager(inputs, targets)      # warm-up: allocations, workspace growth
            g = self._capture(key, inputs, targets)
        for dst, src in zip(g['inputs'] + g['targets'], inputs + targets):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        if not self._grads_clean:
            self._flat_grads.zero_()                            # an eager step / manual backward ran since the last replay
        self.optimizer.sync_hyper(g['grad_scale'])             # lr / betas changed since the last replay? (20-byte copy)
        g['fwd_bwd'].replay()
        if g['update'] is not None:
            parallel.allreduce_gradients(self._flat_grads)      # between the two graphs (scale is baked into 'update')
            g['update'].replay()
        self._grads_clean = True                                # the optimizer launch also cleared the gradient buffer
        return g['stats']

    def _capture(self, key, inputs, targets):
        if self.static_batch_buffers:
            static_in, static_tg = list(inputs), list(targets)
        else:
            static_in = [torch.empty_like(t).copy_(t) for t in inputs]
            static_tg = [torch.empty_like(t).copy_(t) for t in targets]
        self.optimizer._ensure_state(self._flat_params)
        grad_scale = 1.0 / self._world
        self.optimizer.sync_hyper(grad_scale)                   # the captured Adam launch reads them from device memory
        torch.cuda.synchronize()
        g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        # the gradient buffer is cleared by the optimizer launch of the previous replay (DLWPCS_ADAM_ZERO_GRAD), once
        # here for the first one: the step graph needs no fill launch
        self._flat_grads.zero_()
        self._grads_clean = True
        with torch.cuda.graph(g1):
            stats = self._loss_and_backward(static_in, static_tg, True)
            if self._world == 1:        # no exchange step: the update rides in the same graph (no inter-graph gap)
                self.optimizer.apply(self._flat_params, self._flat_grads, grad_scale=grad_scale, zero_grads=True)
                g2 = None
        if g2 is not None:
            with torch.cuda.graph(g2, pool=g1.pool()):
                self.optimizer.apply(self._flat_params, self._flat_grads, grad_scale=grad_scale, zero_grads=True)
        entry = {'fwd_bwd': g1, 'update': g2, 'inputs': static_in, 'targets': static_tg, 'stats': stats,
                 'grad_scale': grad_scale}
        self._graphs[key] = entry
        return entry

    # -------------------------------------------------------------------------------------------------------------- #
    # data plumbing
    # -------------------------------------------------------------------------------------------------------------- #
    def _to_device(self, arr, target=False):
        """host array / tensor -> device tensor in the model's compute dtype (targets stay fp32: the loss is fp32)."""
        dt = torch.float32 if target else backend.torch_dtype(self.compute_dtype)
        if isinstance(arr, torch.Tensor):
            return arr.to(backend.device(), dtype=dt)
        a = np.ascontiguousarray(arr, dtype=np.float32)
        t = torch.from_numpy(a).to(backend.device(), non_blocking=False)
        return t if dt == torch.float32 else t.to(dt)

    def _standardize_inputs(self, x):
        if isinstance(x, dict):
            try:
                return [x[t.layer.name] for t in self.inputs]
            except KeyError as e:
                raise ValueError('No data provided for "%s". Need data for each key in: %s'
                                 % (e.args[0], [t.layer.name for t in self.inputs]))
        xs = _as_list(x)
        if len(xs) != len(self.inputs):
            raise ValueError('Error when checking model input: the list of Numpy arrays that you are passing to your '
                             'model is not the size the model expected. Expected to see %d array(s), but instead got '
                             'the following list of %d arrays' % (len(self.inputs), len(xs)))
        return xs

    def _standardize_targets(self, y):
        if isinstance(y, dict):
            return [y[k] for k in y]
        return _as_list(y)

    def _check_shapes(self, arrays, symbolic, what):
        for a, t in zip(arrays, symbolic):
            if tuple(a.shape[1:]) != tuple(t.shape[1:]):
                raise ValueError('Error when checking %s: expected %s to have shape %s but got array with shape %s'
                                 % (what, t.name, t.shape, tuple(a.shape)))

    def _batches_from(self, x, y, batch_size, shuffle):
        """yield (inputs, targets) lists of host arrays / tensors, one batch at a time."""
        if y is None and hasattr(x, '__getitem__') and hasattr(x, '__len__') and not isinstance(
                x, (np.ndarray, list, tuple, dict, torch.Tensor)):
            for i in range(len(x)):       # keras.utils.Sequence-like (e.g. ArrayDataGenerator)
                item = x[i]
                yield self._standardize_inputs(item[0]), self._standardize_targets(item[1])
            return
        if y is None and hasattr(x, '__iter__') and not isinstance(x, (np.ndarray, list, tuple, dict, torch.Tensor)):
            for item in x:                # dataset-like iterable of (inputs, targets)
                yield self._standardize_inputs(item[0]), self._standardize_targets(item[1])
            return
        xs, ys = self._standardize_inputs(x), self._standardize_targets(y)
        n = xs[0].shape[0]
        bs = n if batch_size is None else int(batch_size)
        if batch_size is None:
            bs = min(32, n)
        idx = np.arange(n)
        if shuffle:
            np.random.shuffle(idx)
        for s in range(0, n, bs):
            sel = idx[s:s + bs]
            if shuffle:
                yield [a[sel] for a in xs], [a[sel] for a in ys]
            else:
                yield [a[s:s + bs] for a in xs], [a[s:s + bs] for a in ys]

    def _n_batches(self, x, y, batch_size):
        if y is None and hasattr(x, '__len__') and not isinstance(x, (np.ndarray, list, tuple, dict, torch.Tensor)):
            return len(x)
        if y is None:
            return None
        n = self._standardize_inputs(x)[0].shape[0]
        bs = min(32, n) if batch_size is None else int(batch_size)
        return -(-n // bs)

    def fit(self, x=None, y=None, batch_size=None, epochs=1, verbose=1, callbacks=None, validation_data=None,
            shuffle=True, initial_epoch=0, steps_per_epoch=None, validation_steps=None, **kwargs):
        if not self._compiled:
            raise RuntimeError('You must compile your model before training/testing. Use `model.compile(...)`.')
        history = cbks.History()
        names = self._metric_names()
        params = {'epochs': epochs, 'verbose': verbose, 'metrics': names, 'steps': self._n_batches(x, y, batch_size)}
        cbl = cbks.CallbackList([history] + list(callbacks or []), self, params)
        self.history = history
        self.stop_training = False
        cbl.call('on_train_begin', None)
        dev = backend.device()
        for epoch in range(initial_epoch, epochs):
            if self.stop_training:
                break
            t0 = time.time()
            cbl.call('on_epoch_begin', epoch, None)
            sums = torch.zeros((len(self.outputs), 2), dtype=torch.float32, device=dev)
            count = 0
            for bi, (bx, by) in enumerate(self._batches_from(x, y, batch_size, shuffle)):
                if steps_per_epoch is not None and bi >= steps_per_epoch:
                    break
                if cbl.wants_batch_logs:
                    cbl.call('on_train_batch_begin', bi, None)
                dx = [self._to_device(a) for a in bx]
                dt = [self._to_device(a, target=True) for a in by]
                if count == 0 and epoch == initial_epoch:
                    self._check_shapes(dx, self.inputs, 'input')
                    self._check_shapes(dt, self.outputs, 'target')
                stats = self.train_on_device_batch(dx, dt)
                sums += stats
                count += 1
                if cbl.wants_batch_logs:
                    vals = self._assemble_logs(stats, 1)
                    cbl.call('on_train_batch_end', bi, dict(zip(names, vals), batch=bi, size=int(dx[0].shape[0])))
            logs = dict(zip(names, self._assemble_logs(sums, count)))
            if validation_data is not None:
                vals = self._evaluate_impl(validation_data, None, batch_size, validation_steps)
                logs.update({'val_' + k: v for k, v in zip(names, vals)})
            cbl.call('on_epoch_end', epoch, logs)
            if y is None and hasattr(x, 'on_epoch_end'):
                x.on_epoch_end()           # keras.utils.Sequence contract (re-shuffles an ArrayDataGenerator)
            if verbose:
                msg = ' - '.join('%s: %.4f' % (k, v) for k, v in logs.items())
                print('Epoch %d/%d - %ds - %s' % (epoch + 1, epochs, int(time.time() - t0), msg))
        cbl.call('on_train_end', None)
        return history

    def fit_generator(self, generator, **kwargs):
        return self.fit(generator, **kwargs)

    def _evaluate_impl(self, x, y, batch_size, steps):
        if isinstance(x, tuple) and y is None and len(x) in (2, 3) and not hasattr(x, 'shape'):
            x, y = x[0], x[1]
        dev = backend.device()
        sums = torch.zeros((len(self.outputs), 2), dtype=torch.float64, device=dev)
        total = 0
        with torch.no_grad():
            for bi, (bx, by) in enumerate(self._batches_from(x, y, batch_size, False)):
                if steps is not None and bi >= steps:
                    break
                dx = [self._to_device(a) for a in bx]
                dt = [self._to_device(a, target=True) for a in by]
                n = dx[0].shape[0]
                stats = self._loss_and_backward(dx, dt, train=False)
                sums += stats.double() * n            # keras weights batches by their size
                total += n
        return self._assemble_logs((sums / max(total, 1)).float(), 1)

    def evaluate(self, x=None, y=None, batch_size=None, verbose=1, steps=None, **kwargs):
        if not self._compiled:
            raise RuntimeError('You must compile your model before training/testing. Use `model.compile(...)`.')
        vals = self._evaluate_impl(x, y, batch_size, steps)
        return vals[0] if len(vals) == 1 else vals

    def predict(self, x, batch_size=None, verbose=0, steps=None, **kwargs):
        xs = self._standardize_inputs(x)
        n = xs[0].shape[0]
        bs = 32 if batch_size is None else int(batch_size)
        outs = None
        with torch.no_grad():
            for s in range(0, n, bs):
                dx = [self._to_device(a[s:s + bs]) for a in xs]
                if s == 0:
                    self._check_shapes(dx, self.inputs, 'input')
                res = self._forward(dx, repack=(s == 0))
                if outs is None:
                    outs = [np.empty((n,) + tuple(r.shape[1:]), dtype=np.float32) for r in res]
                for o, r in zip(outs, res):
                    o[s:s + bs] = r.float().cpu().numpy()
        if outs is None:
            outs = [np.empty((0,) + tuple(o.shape[1:]), dtype=np.float32) for o in self.outputs]
        return outs[0] if self._single_output else outs

    def predict_on_device(self, inputs, repack=True):
        """Forward pass on device tensors without host round trips (used by the device-resident rollout).
        repack=False skips the weight-packing launch: only for consecutive passes with unchanged weights."""
        with torch.no_grad():
            outs = self._forward(_as_list(inputs), repack=repack)
        return outs[0] if self._single_output else outs

    def rollout_on_device(self, predictors, steps, n_steps, out_series, verbose=0, batch_size=None):
        """
        Autoregressive rollout with the state resident in HBM (replaces the per-step numpy round trip of the reference's
        predict_timeseries loop, DLWP/model/models.py:446-454).  Fills out_series[(steps*n_steps), n, ...] in place.
        """
        if len(self.inputs) != 1:
            raise NotImplementedError('rollout_on_device needs a single-input model')
        n = predictors.shape[0]
        bs = 32 if batch_size is None else int(batch_size)
        with torch.no_grad():
            for s in range(0, n, bs):
                state = self._to_device(predictors[s:s + bs])
                self._check_shapes([state], self.inputs, 'input')
                for t in range(steps):
                    if verbose > 0 and s == 0:
                        print('Prediction step %d/%d' % (t + 1, steps))
                    res = self._forward([state], repack=(t == 0 and s == 0))       # weights are fixed during a rollout
                    if tuple(res[-1].shape) != tuple(state.shape):
                        raise ValueError('could not broadcast model output of shape %s into the input of shape %s'
                                         % (tuple(res[-1].shape), tuple(state.shape)))
                    state = res[-1]
                    for k in range(n_steps):
                        out_series[t * n_steps + k, s:s + bs] = res[k].float().cpu().numpy()

    def reset_states(self):
        pass

    # -------------------------------------------------------------------------------------------------------------- #
    # persistence (native format; HDF5 needs h5py which this stack does not carry)
    # -------------------------------------------------------------------------------------------------------------- #
    def save_weights(self, filepath, overwrite=True, save_format=None):
        if not overwrite and os.path.exists(filepath):
            return
        payload = {'format': 'dlwpcs-weights-1', 'names': [n for l in self._weight_layers() for n in l._weight_names],
                   'weights': self.get_weights()}
        tmp = '%s.tmp%d' % (filepath, os.getpid())
        with open(tmp, 'wb') as f:
            pickle.dump(payload, f, protocol=pickle.HIGHEST_PROTOCOL)
        os.replace(tmp, filepath)

    def load_weights(self, filepath, by_name=False):
        with open(filepath, 'rb') as f:
            payload = pickle.load(f)
        if not isinstance(payload, dict) or payload.get('format') != 'dlwpcs-weights-1':
            raise ValueError('%s is not a dlwpcs weights file' % filepath)
        self.set_weights(payload['weights'])

    def get_config(self):
        index = {}
        layer_cfgs = []
        for l in self.layers:
            index[id(l)] = len(layer_cfgs)
            layer_cfgs.append({'class_name': type(l).__name__, 'config': l.get_config()})
        tid = {}
        nodes = []
        for t in self.inputs:
            tid[t.uid] = len(nodes)
            nodes.append({'layer': index[id(t.layer)], 'inputs': []})
        for t in self._nodes:
            if t.uid in tid:
                continue
            tid[t.uid] = len(nodes)
            nodes.append({'layer': index[id(t.layer)], 'inputs': [tid[i.uid] for i in t.node_inputs]})
        return {'name': self.name, 'layers': layer_cfgs, 'nodes': nodes,
                'inputs': [tid[t.uid] for t in self.inputs], 'outputs': [tid[t.uid] for t in self.outputs],
                'single_input': self._single_input, 'single_output': self._single_output}

    @classmethod
    def from_config(cls, config, custom_objects=None):
        from .. import custom
        from . import layers as klayers
        table = {}
        for mod in (klayers, custom):
            for k in dir(mod):
                v = getattr(mod, k)
                if isinstance(v, type) and issubclass(v, Layer):
                    table[k] = v
        table.update(custom_objects or {})
        layers = []
        for lc in config['layers']:
            if lc['class_name'] not in table:
                raise ValueError('Unknown layer: %s' % lc['class_name'])
            cfg = dict(lc['config'])
            layers.append(table[lc['class_name']].from_config(cfg))
        tensors = []
        for nd in config['nodes']:
            lay = layers[nd['layer']]
            if isinstance(lay, InputLayer):
                tensors.append(KTensor(lay.batch_input_shape, layer=lay, node_inputs=(), name=lay.name))
            else:
                ins = [tensors[i] for i in nd['inputs']]
                tensors.append(lay(ins if (isinstance(lay, Concatenate) or len(ins) > 1) else ins[0]))
        ins = [tensors[i] for i in config['inputs']]
        outs = [tensors[i] for i in config['outputs']]
        return cls(inputs=ins[0] if config.get('single_input') else ins,
                   outputs=outs[0] if config.get('single_output') else outs, name=config.get('name'))

    def save(self, filepath, overwrite=True, include_optimizer=True, **kwargs):
        if not overwrite and os.path.exists(filepath):
            return
        payload = {'format': 'dlwpcs-model-1', 'config': self.get_config(), 'weights': self.get_weights(),
                   'compile': None, 'compute_dtype': self.compute_dtype}
        if self._compiled:
            payload['compile'] = {'loss': self.loss, 'loss_weights': self.loss_weights, 'metrics': self.metrics,
                                  'optimizer': self.optimizer.get_config(),
                                  'optimizer_state': self.optimizer.state_dict() if include_optimizer else None}
        tmp = '%s.tmp%d' % (filepath, os.getpid())
        with open(tmp, 'wb') as f:
            pickle.dump(payload, f, protocol=pickle.HIGHEST_PROTOCOL)
        os.replace(tmp, filepath)

    def to_json(self, **kwargs):
        return json.dumps({'class_name': 'Model', 'config': self.get_config()}, default=lambda o: list(o), **kwargs)

    def summary(self, line_length=None, positions=None, print_fn=None):
        print_fn = print_fn or print
        print_fn('Model: "%s"' % self.name)
        print_fn('%-34s %-28s %10s' % ('Layer (type)', 'Output Shape', 'Param #'))
        print_fn('=' * 74)
        shapes = {}
        for t in self.inputs + self._nodes:
            shapes.setdefault(id(t.layer), t.shape)
        for l in self.layers:
            print_fn('%-34s %-28s %10d' % ('%s (%s)' % (l.name, type(l).__name__), str(shapes.get(id(l))),
                                           l.count_params()))
        print_fn('=' * 74)
        print_fn('Total params: {:,}'.format(self.count_params()))
        print_fn('Fused cubed-sphere convolution launches per forward pass: %d' % self.n_fused)


def load_model(filepath, custom_objects=None, compile=True):
    with open(filepath, 'rb') as f:
        payload = pickle.load(f)
    if not isinstance(payload, dict) or payload.get('format') != 'dlwpcs-model-1':
        raise ValueError('%s is not a dlwpcs model file (HDF5 models written by TensorFlow need h5py + TF to convert)'
                         % filepath)
    model = Model.from_config(payload['config'], custom_objects=custom_objects)
    model.compute_dtype = payload.get('compute_dtype', model.compute_dtype)      # the mixed-precision mode travels with it
    model.set_weights(payload['weights'])
    cmp = payload.get('compile')
    if compile and cmp:
        model.compile(optimizer=optimizers.get(cmp['optimizer']), loss=cmp['loss'], loss_weights=cmp['loss_weights'],
                      metrics=cmp['metrics'])
        model.optimizer.load_state_dict(cmp.get('optimizer_state'), model._flat_params)
    return model


def clone_model(model):
    new = Model.from_config(model.get_config())
    new.compute_dtype = model.compute_dtype
    new.set_weights(model.get_weights())
    return new
