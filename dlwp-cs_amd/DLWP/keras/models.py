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
import gc
import json
import os
import sys
import time

import numpy as np
import torch

from .. import ops, parallel
from ..options import option
from .. import _native as _nat
from .._native import ACT_LEAKY_CLIP, ACT_NONE
from . import backend, callbacks as cbks, optimizers, staging
from .engine import KTensor, Layer
from .layers import AveragePooling3D, Concatenate, InputLayer, ReLU, UpSampling3D


def _new_graph():
    """DLWPCS_KEEP_GRAPH=1 (bench.py's launch census): the hipGraph object is kept so that its nodes can be counted."""
    if os.environ.get('DLWPCS_KEEP_GRAPH', '0') == '1':
        return torch.cuda.CUDAGraph(keep_graph=True)
    return torch.cuda.CUDAGraph()


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
        self.use_graphs = option('graphs')
        # activation dtype on the device ('float32' | 'bfloat16'); parameters / gradients / Adam state are always fp32
        self.compute_dtype = backend.compute_dtype()
        self.prepack_weights = True
        self.wgrad_side_stream = False     # (weight gradients on a second stream: measured slower on MI355X)
        # one reduction launch for all layers' weight-gradient partials (DLWPCS_CONV_DEFER_REDUCE)
        self.defer_wgrad_reduce = True
        # one persistent launch for the weight gradients of all layers of a step (ops.wgrad_batch)
        self.batch_wgrad = option('wgrad_batch')
        self.check_finite = option('check_finite')
        # the optimizer inside the reduction of the batched weight gradients (dlwpcs_wgrad_batch_adam; world size 1)
        self.fuse_adam = option('fuse_adam')
        # ring fix-up of a pooled tensor's gradient inside the pooling adjoint (dlwpcs_avgpool2_bwd_ring)
        self.fold_ring = option('fold_ring')
        # 2x2 average pooling written by the epilogue of the convolution in front of it (dlwpcs_conv_fwd_pool)
        self.fuse_pool = option('fuse_pool')
        # second stage of the fused head's loss reduction inside the step's last launch (dlwpcs_wgrad_batch_adam_tail)
        self.fold_loss_tail = option('fold_loss_tail')
        # hipGraph-replayed steps: the reduction + optimizer launch also refreshes the packed bf16 operands, the step starts
        # without the packing launch (dlwpcs_wgrad_batch_adam_tail with pack items)
        self.fuse_pack = option('fuse_pack')
        self._packed_ok = False             # the packed operands hold the current parameters (see _ensure_packed)
        # data-parallel exchange in two buckets (exchange_buckets = 2 / DLWPCS_EXCHANGE_BUCKETS=2): the gradients of the
        # decoder-side layers are summed over the ranks WHILE the encoder-side half of the backward pass runs.  Default 1: one
        # all-reduce behind the whole backward pass -- whether the overlap pays depends on how RCCL's workgroups share the
        # CUs with the persistent one-workgroup-per-CU kernels, which only a multi-GPU node can show; bench.py times both
        # at N > 1 and runs the faster.  (The split step also runs at world size 1, for the tests.)
        self.exchange_buckets = int(os.environ.get('DLWPCS_EXCHANGE_BUCKETS', '1'))
        self._stager = None                 # pinned-memory / copy-stream feed of fit() on host arrays (keras/staging.py)
        # training step: output layer + loss + loss gradient + the layer's data gradient as one launch (ops.head_mse)
        self.fuse_head_loss = option('fuse_head')
        # inference: the pointwise output layer inside the epilogue of the convolution in front of it (ops.cs_conv_head)
        self.fold_head = option('fold_head')
        # True: the caller feeds every step through the SAME device tensors (e.g. a generator that assembles each batch in
        # place): the captured graphs read them directly instead of copying each batch into private static buffers
        self.static_batch_buffers = False
        self._ones = {}
        self._compiled = False
        self._flat_params = self._flat_grads = None
        self._grads_clean = False
        self._graphs = {}
        self._infer_graphs = {}         # rollout_passes_on_device: captured chains of forward passes
        self._rollout_series = {}       # rollout_on_device: the device buffer a batch's series is written to
        self.rollout_keep_bytes = 4 << 30   # ... kept between calls (with its captured chain) up to this size
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
        # A graph whose every layer is channels_first (reference DLWP/custom.py:1085-1196: the layers' default) runs
        # channels_last INSIDE the plan -- the layout of the kernels -- between one transpose at the inputs and one at the
        # outputs, instead of two transposes around every layer.  Conditions: rank-5 tensors, only layer kinds whose
        # channels_last twin the plan knows, concatenation along the channel axis (1).
        def cf_layer(t):
            lay = t.layer
            if isinstance(lay, InputLayer):
                return len(t.shape) == 5
            if isinstance(lay, CubeSphereConv2D) and lay.activation is not None \
                    and getattr(lay.activation, '_dlwp_name', None) != 'relu':
                return False        # a user's callable may depend on the axis order (softmax): such a graph keeps its own layout
            if isinstance(lay, (CubeSpherePadding2D, CubeSphereConv2D, AveragePooling3D, UpSampling3D)):
                return lay.data_format == 'channels_first'
            if isinstance(lay, Concatenate):
                return lay._axis(len(t.shape)) == 1
            return isinstance(lay, ReLU)
        self._cf_model = (option('cf_model') and all(cf_layer(t) for t in self._nodes)
                          and any(isinstance(t.layer, (CubeSphereConv2D, CubeSpherePadding2D)) for t in self._nodes))
        fmt = 'channels_first' if self._cf_model else 'channels_last'
        virtual = set()          # tensors never materialised
        fused = {}               # uid of the tensor a fused step produces -> step description
        for t in self._nodes:
            lay = t.layer
            if not isinstance(lay, CubeSphereConv2D) or len(t.node_inputs) != 1:
                continue
            padt = t.node_inputs[0]
            if not isinstance(padt.layer, CubeSpherePadding2D) or sole_consumer(padt) is not t:
                continue
            if padt.layer.data_format != fmt or lay.data_format != fmt or not lay.can_fuse_halo(padt.layer.padding[1][0]):
                continue
            if padt.layer.padding[1] != padt.layer.padding[2] or padt.layer.padding[1][0] != padt.layer.padding[1][1]:
                continue
            x = padt.node_inputs[0]
            src0, src1, up0 = x, None, False
            chain = [padt]
            if isinstance(x.layer, Concatenate) and len(x.node_inputs) == 2 and sole_consumer(x) is padt \
                    and x.layer._axis(len(x.shape)) == (1 if self._cf_model else len(x.shape) - 1):
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
            elif isinstance(t.layer, AveragePooling3D) and len(t.node_inputs) == 1 and self._runs_channels_last(t.layer) and (
                    len(consumers.get(t.node_inputs[0].uid, [])) > 1 or t.node_inputs[0].uid in out_uids):
                # the pooled tensor has other consumers (U-Net skip connection): they are re-routed through an alias so
                # that both gradients reach ONE backward kernel (ops.avgpool2_skip)
                steps.append(('pool_skip', t.uid, t.layer, t.node_inputs[0].uid))
            else:
                steps.append(('layer', t.uid, t.layer, [i.uid for i in t.node_inputs],
                              isinstance(t.layer, Concatenate)))
        self._plan = steps
        self.n_fused = len(fused)
        # padded rollout state (rollout_on_device: the output layer writes ceil8(C) channels, that tensor comes back as the
        # input): only when the model's single input is read by ONE step, a fused convolution taking it as its plain source 0
        self._padded_io_ok = False
        if len(self.inputs) == 1:
            u = self.inputs[0].uid
            rd = [st for st in steps
                  if (st[0] == 'fused_conv' and u in (st[3], st[4])) or (st[0] == 'pool_skip' and st[3] == u)
                  or (st[0] == 'layer' and u in st[3])]
            self._padded_io_ok = (len(rd) == 1 and rd[0][0] == 'fused_conv' and rd[0][3] == u and rd[0][4] is None
                                  and not rd[0][5] and u not in out_uids and not self._cf_model)
        self._plan_premask(steps, out_uids)
        # inference: a fused convolution whose only reader is a pointwise CubeSphereConv2D (the U-Net's output layer,
        # Azure/train_cs.py:300-305) can take that layer into its epilogue (ops.cs_conv_head): step index -> the head's step index
        self._head_fold = {}
        for i, st in enumerate(steps):
            if st[0] != 'fused_conv' or st[1] in out_uids:
                continue
            rd = [j for j, s2 in enumerate(steps)
                  if (s2[0] == 'fused_conv' and st[1] in (s2[3], s2[4])) or (s2[0] == 'pool_skip' and s2[3] == st[1])
                  or (s2[0] == 'layer' and st[1] in s2[3])]
            if len(rd) != 1 or steps[rd[0]][0] != 'layer':
                continue
            hl = steps[rd[0]][2]
            if (isinstance(hl, CubeSphereConv2D) and len(steps[rd[0]][3]) == 1 and tuple(hl.kernel_size) == (1, 1)
                    and hl.activation is None and not hl.independent_north_pole and not st[2].independent_north_pole
                    and st[2].filters == 32 and self._runs_channels_last(hl) and not self._cf_model):
                self._head_fold[i] = rd[0]
        # model outputs that no other node consumes (candidates for the fused head + loss step) and appear once
        uids = [o.uid for o in self.outputs]
        self._sole_outputs = {u for u in uids if not consumers.get(u) and uids.count(u) == 1}

    def _plan_premask(self, steps, out_uids):
        """Pre-masked gradient convention (include/dlwpcs.h, dlwpcs_conv_bwd_data_masked): the output y of a fused
        convolution + ReLU receives its gradient already multiplied by act'(y) when EVERY consumer of y is an op of this engine
        that can apply the mask where it produces the gradient -- another fused convolution (data-gradient epilogue / routing
        kernels), the pooling node of a skip connection, the pointwise output layer.  Decided statically, per step:
        self._premask[uid] = (negative_slope, max_value) of the tensors that carry the convention, and per consuming step
        which of its sources it must mask (a consumer that runs after the tensor's 'pool_skip' step sees the ALIAS, which
        hands plain gradients back to the pooling node)."""
        from ..custom import CubeSphereConv2D
        produced = {}                  # uid -> (alpha, vmax) of fused conv + activation outputs
        for st in steps:
            if st[0] == 'fused_conv' and st[6] == ACT_LEAKY_CLIP:
                produced[st[1]] = (float(st[7]), float(st[8]))
        capable = {u: (u not in out_uids) for u in produced}
        # a layer that computes no data gradient (its sources are model inputs) is the only reader of its dz: the batched
        # weight-gradient kernel applies act' on load there, cheaper than a masking epilogue in the consumer's data gradient
        in_uids_model = {t.uid for t in self.inputs}
        for st in steps:
            if st[0] == 'fused_conv' and st[1] in capable and st[3] in in_uids_model and (st[4] is None or st[4] in in_uids_model):
                capable[st[1]] = False
        aliased = set()
        src_mask = {}                  # step index -> (mask source 0?, mask source 1?) | bool for pool / head steps
        for i, st in enumerate(steps):
            if st[0] == 'fused_conv':
                s0, s1 = st[3], st[4]
                src_mask[i] = (s0 in produced and s0 not in aliased, s1 is not None and s1 in produced and s1 not in aliased)
                if s0 == s1 and s0 in produced:
                    capable[s0] = False            # (both sources the same tensor: keep the plain path)
            elif st[0] == 'pool_skip':
                u = st[3]
                src_mask[i] = u in produced and u not in aliased
                aliased.add(u)
            else:
                _, out_uid, lay, in_uids, _ = st
                head = (isinstance(lay, CubeSphereConv2D) and len(in_uids) == 1 and lay._is_mfma_config()
                        and self._runs_channels_last(lay) and lay.activation is None)
                src_mask[i] = bool(head and in_uids[0] in produced and in_uids[0] not in aliased)
                for u in in_uids:
                    if u in produced and u not in aliased and not head:
                        capable[u] = False         # a consumer that cannot mask: the tensor keeps plain gradients
        self._premask = {u: produced[u] for u in produced if capable[u]}
        self._src_mask = src_mask
        # Ring fix-up folded into the pooling adjoint: the pooled tensor of a masked 'pool_skip' step whose ONLY reader is one
        # fused convolution (as its plain source 0) -- that convolution's data gradient leaves the halo ring to the pooling node
        readers = {}
        for i, st in enumerate(steps):
            ins = [st[3], st[4]] if st[0] == 'fused_conv' else ([st[3]] if st[0] == 'pool_skip' else list(st[3]))
            for u in ins:
                if u is not None:
                    readers.setdefault(u, []).append(i)
        # 2x2 pooling as a second output of the producing convolution (dlwpcs_conv_fwd_pool): the fused convolutions whose
        # output's NEXT reader is a 'pool_skip' step (the by-product is taken by that step right away, see ops._POOLED)
        self._pool_producers = set()
        for i, st in enumerate(steps):
            if st[0] == 'fused_conv':
                r = readers.get(st[1], [])
                if r and steps[r[0]][0] == 'pool_skip' and steps[r[0]][3] == st[1]:
                    self._pool_producers.add(i)
        self._defer_ring = set()
        for i, st in enumerate(steps):
            if st[0] != 'pool_skip' or not (src_mask.get(i) and st[3] in self._premask):
                continue
            r = readers.get(st[1], [])
            if len(r) == 1 and st[1] not in out_uids and steps[r[0]][0] == 'fused_conv' and steps[r[0]][3] == st[1] \
                    and steps[r[0]][4] != st[1] and not steps[r[0]][5]:
                self._defer_ring.add(r[0])

    def _runs_channels_last(self, lay):
        return lay.data_format == 'channels_last' or self._cf_model

    def _plan_exchange(self):
        """Two-bucket exchange: (first step of bucket A, element offset of bucket A in the flat gradient buffer) or None.
        Bucket A = the weights of the steps from the cut on (the decoder side: their gradients are final first), bucket B the
        rest; the cut is the latest step from which on at least 40 % of the parameters lie.  The flat buffers are in layer
        (= execution) order, so each bucket is one contiguous slice -- checked, None if a model breaks that."""
        if self._exchange_cut is not None:
            return self._exchange_cut or None
        self._exchange_cut = False
        if self._flat_grads is None:
            return None
        base = self._flat_grads.data_ptr()
        spans = []                      # per step: (lo, hi) element range of its layer's gradients in the flat buffer
        for st in self._plan:
            ws = [w for w in getattr(st[2], '_weights', []) if w.requires_grad and w.grad is not None]
            if ws:
                lo = min((w.grad.data_ptr() - base) // 4 for w in ws)
                hi = max((w.grad.data_ptr() - base) // 4 + w.numel() for w in ws)
                spans.append((lo, hi))
            else:
                spans.append(None)
        total = sum(hi - lo for sp in spans if sp for lo, hi in [sp])
        if total == 0:
            return None
        acc, cut = 0, None
        for i in range(len(spans) - 1, 0, -1):
            if spans[i]:
                acc += spans[i][1] - spans[i][0]
                if acc >= 0.4 * total:
                    cut = i
                    break
        if cut is None or not any(spans[:cut]):
            return None
        lo_a = min(sp[0] for sp in spans[cut:] if sp)
        if any(sp and sp[1] > lo_a for sp in spans[:cut]):
            return None                 # a layer applied on both sides, or an unusual layer order: single bucket
        self._exchange_cut = (cut, int(lo_a) // 64 * 64)
        return self._exchange_cut

    def _premask_on(self):
        return (self.compute_dtype == 'bfloat16' and torch.is_grad_enabled()
                and option('premask'))

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
        entries, table, by_grad = [], {}, {}
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
            if we.grad is not None:
                by_grad[we.grad.data_ptr()] = entries[-1]
        st = {'key': key, 'items': ops.make_pack_items(entries, device) if entries else None, 'n': len(entries),
              'table': table, 'keep': entries, 'by_grad': by_grad}
        self._pack_cache = st
        return st

    def _forward(self, inputs, repack=True, fuse_targets=None):
        want = backend.torch_dtype(self.compute_dtype)
        inputs = [v if v.dtype == want else v.to(want) for v in inputs]
        if inputs and inputs[0].is_cuda and self.prepack_weights:
            # one launch packs the weights of every layer for this pass (they changed with the last optimizer step);
            # repack=False: the caller knows they have not changed since its previous pass (steps of one rollout)
            st = self._pack_state(inputs[0].device)
            if st['n'] and (repack or not st.get('packed')):
                ops.pack_batch(st['items'], st['n'])
                st['packed'] = True
                st['packed_version'] = self._flat_params._version if self._flat_params is not None else None
                self._packed_ok = True
            ops.PREPACKED = st['table']
            try:
                return self._run_plan(inputs, fuse_targets)
            finally:
                ops.PREPACKED = {}
        return self._run_plan(inputs, fuse_targets)

    def _run_plan(self, inputs, fuse_targets=None):
        """fuse_targets (training step only): {output tensor uid: (target, loss weight)} -- an output produced by a pointwise
        CubeSphereConv2D that nothing else consumes is then computed together with its loss, its loss gradient and the layer's
        data gradient by ONE launch (ops.head_mse); the entry of the returned list is the (2,) stats tensor instead of the
        prediction."""
        from ..custom import CubeSphereConv2D, CubeSpherePadding2D
        cf = self._cf_model
        if cf:
            inputs = [ops.channels_first_to_last(v) for v in inputs]        # ONE transpose per input (and per output, below)
        values = {t.uid: v for t, v in zip(self.inputs, inputs)}
        self._fused_outputs = set()         # uids whose entry of the returned list is the (2,) stats tensor of ops.head_mse
        pm = self._premask if self._premask_on() else {}
        cut = getattr(self, '_record_cut', None)
        self._cut_tensors = []
        ops._POOLED.clear()
        folded = set()                       # head steps served by the epilogue of the convolution in front of them
        for i, st in enumerate(self._plan):
            if i in folded:
                continue
            if cut is not None and i == cut:
                # split backward pass (two-bucket exchange): the tensors alive here that a step from here on READS -- the
                # first half of the backward pass stops at them (an aliased skip tensor is its alias by now)
                need, seen = set(), set()
                for s2 in self._plan[cut:]:
                    ins = [s2[3], s2[4]] if s2[0] == 'fused_conv' else ([s2[3]] if s2[0] == 'pool_skip' else list(s2[3]))
                    need.update(u for u in ins if u is not None)
                for u, v in values.items():
                    if u in need and isinstance(v, torch.Tensor) and v.requires_grad and id(v) not in seen:
                        seen.add(id(v))
                        self._cut_tensors.append(v)
            if st[0] == 'fused_conv':
                _, out_uid, lay, s0, s1, up0, act, alpha, vmax = st
                j = self._head_fold.get(i) if self.fold_head else None
                if j is not None and fuse_targets is None and not torch.is_grad_enabled():
                    hl, h_uid = self._plan[j][2], self._plan[j][1]
                    padded = bool(getattr(self, '_padded_io', False) and h_uid in self._sole_outputs and hl.filters % 8 != 0)
                    if (lay._is_mfma_config() and hl._is_mfma_config() and hl.built and lay.built and ops.cs_conv_head_applicable(
                            values[s0], None if s1 is None else values[s1], lay.equatorial_kernel, hl.equatorial_kernel, padded)):
                        values[h_uid] = ops.cs_conv_head(values[s0], lay.equatorial_kernel, lay.equatorial_bias,
                                                         hl.equatorial_kernel, hl.equatorial_bias,
                                                         src1=None if s1 is None else values[s1], up0=up0,
                                                         flip_north_pole=lay.flip_north_pole, act=act, alpha=alpha, vmax=vmax,
                                                         out_padded=padded)
                        folded.add(j)
                        continue
                m0, m1 = self._src_mask[i]
                values[out_uid] = lay.fused_call(values[s0], None if s1 is None else values[s1], up0=up0, halo=True,
                                                 act=act, alpha=alpha, vmax=vmax,
                                                 premask0=pm.get(s0) if m0 else None, premask1=pm.get(s1) if m1 else None,
                                                 dy_premasked=out_uid in pm,
                                                 defer_ring0=bool(pm) and self.fold_ring and i in self._defer_ring,
                                                 want_pool=self.fuse_pool and i in self._pool_producers)
            elif st[0] == 'pool_skip':
                _, out_uid, lay, in_uid = st
                values[out_uid], values[in_uid] = ops.avgpool2_skip(values[in_uid],
                                                                     pm.get(in_uid) if self._src_mask[i] else None)
                if cut is not None and i < cut:
                    values[in_uid] = ops.detour(values[in_uid])     # split backward pass: see ops._Detour
            else:
                _, out_uid, lay, in_uids, takes_list = st
                args = [values[u] for u in in_uids]
                head_pm = pm.get(in_uids[0]) if (self._src_mask[i] and in_uids) else None
                if (fuse_targets is not None and out_uid in fuse_targets and out_uid in self._sole_outputs
                        and isinstance(lay, CubeSphereConv2D) and len(args) == 1 and lay._is_mfma_config()
                        and self._runs_channels_last(lay) and lay.activation is None and lay.north_pole_kernel is None
                        and ops.head_mse_applicable(args[0], lay.equatorial_kernel, lay.kernel_size[0], ACT_NONE,
                                                    fuse_targets[out_uid][0])):
                    tgt, wgt = fuse_targets[out_uid]
                    values[out_uid] = ops.head_mse(args[0], tgt, lay.equatorial_kernel, lay.polar_kernel, lay.equatorial_bias,
                                                   lay.polar_bias, wgt, lay.flip_north_pole, premask=head_pm)
                    self._fused_outputs.add(out_uid)
                    continue
                if head_pm is not None:
                    # pointwise / 'valid' convolution on a pre-masked source: the layer's own call with the mask handed down
                    values[out_uid] = lay.fused_call(args[0], halo=False, premask0=head_pm)
                    continue
                if (getattr(self, '_padded_io', False) and out_uid in self._sole_outputs and isinstance(lay, CubeSphereConv2D)
                        and len(args) == 1 and lay._is_mfma_config() and self._runs_channels_last(lay)
                        and lay.activation is None and lay.kernel_size[0] == 1 and args[0].dtype == torch.bfloat16
                        and args[0].shape[-1] == 32 and 8 <= lay.filters <= 32 and lay.filters % 2 == 0
                        and (args[0].shape[2] * args[0].shape[3]) % 16 == 0):
                    # rollout: the output layer writes its rows padded to the 16-B vector, the layout the first layer reads fastest
                    values[out_uid] = lay.fused_call(args[0], halo=False, out_padded=True)
                    continue
                if cf:
                    # the channels_last twin of the layer (the graph was checked in _build_plan)
                    if isinstance(lay, CubeSphereConv2D):
                        values[out_uid] = lay.call(args[0], channels_last_io=True)
                    elif isinstance(lay, CubeSpherePadding2D):
                        values[out_uid] = ops.cs_pad(args[0], lay.padding[1][0])
                    elif isinstance(lay, AveragePooling3D):
                        values[out_uid] = ops.avgpool2(args[0])
                    elif isinstance(lay, UpSampling3D):
                        values[out_uid] = ops.upsample2(args[0])
                    elif isinstance(lay, Concatenate):
                        values[out_uid] = ops.concat_channels(list(args))
                    else:
                        values[out_uid] = lay.call(args[0])              # ReLU: layout-agnostic
                    continue
                values[out_uid] = lay.call(args if (takes_list or len(args) > 1) else args[0])
        if cf and not getattr(self, '_loss_in_cl', False):
            return [values[o.uid] if o.uid in self._fused_outputs else ops.channels_last_to_first(values[o.uid])
                    for o in self.outputs]
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
        self._wrules = None                                     # (the weight tensors were re-bound into the flat buffer)
        self._flat_len = off
        self._n_params = total
        self._exchange_cut = None

    # -------------------------------------------------------------------------------------------------------------- #
    # compile / train
    # -------------------------------------------------------------------------------------------------------------- #
    def compile(self, optimizer='adam', loss=None, metrics=None, loss_weights=None, **kwargs):
        self.optimizer = optimizers.get(optimizer)
        mp_dtype = getattr(self.optimizer, '_mixed_precision', None)
        if mp_dtype is not None and mp_dtype != backend.compute_dtype() and mp_dtype != self.compute_dtype:
            # the global policy was changed again after the optimizer was tagged by enable_mixed_precision_graph_rewrite():
            # the tag still wins (the switch travels with the optimizer, like TF's rewrite), but say so
            import warnings
            warnings.warn('compile(): the optimizer was returned by enable_mixed_precision_graph_rewrite(); the model is put '
                          'into the %s mode although the global policy is %s now (pass a fresh optimizer to avoid this)'
                          % (mp_dtype, backend.compute_dtype()))
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
        self._infer_graphs.clear()          # (captured with the parameters' old addresses)
        self._seen_batch.clear()
        self._world = parallel.world()[1]
        parallel.broadcast_parameters(self._flat_params)     # identical replicas: rank 0's initial weights everywhere
        if parallel.exchange_wanted() and self._flat_params.is_cuda:
            parallel.native_comm()      # (opt-in; agreed on by every rank HERE, outside any capture: see DLWP/parallel.py)
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
        vals = [float(s[:, 0].sum())]                           # (a model with weight regularizers: its last row is the penalty)
        s = s[:len(self.outputs)]
        if len(self.outputs) > 1:
            vals += [float(s[i, 0] / w[i]) if w[i] != 0 else 0.0 for i in range(len(self.outputs))]
            if self.metrics:
                vals += [float(s[i, 1]) for i in range(len(self.outputs))]
        elif self.metrics:
            vals += [float(s[0, 1])]
        return vals

    def _loss_and_backward(self, inputs, targets, train=True, fuse_update=None, skip_pack=False, exchange=False):
        """fuse_update = grad_scale | None: with a value, the optimizer step may ride in the reduction of the batched weight
        gradients (ops.flush_wgrad_batch); self._update_done says whether it did (self._pack_done: and refreshed the packed
        operands).  exchange=True (data parallel, with fuse_update): the gradient all-reduce is issued inside the step, between
        the reduction and ONE launch that applies the update (ops.apply_wgrad_batch) -- the whole step is one launch sequence,
        capturable as one hipGraph."""
        gen = self._loss_and_backward_gen(inputs, targets, train, fuse_update, split=False, skip_pack=skip_pack, exchange=exchange)
        try:
            while True:
                next(gen)
        except StopIteration as e:
            return e.value

    def _split_wanted(self):
        """Two-bucket exchange for this step?"""
        return self.exchange_buckets == 2 and self._plan_exchange() is not None

    def _adam_args(self, dev, fuse_update):
        """(adam tuple, pack lookup) for the fused / applied optimizer launch"""
        opt = self.optimizer
        opt._ensure_state(self._flat_params)
        if not torch.cuda.is_current_stream_capturing():
            opt.sync_hyper(fuse_update)
        adam = (self._flat_params, self._flat_grads, opt._m, opt._v, opt._step, opt._hyper, sum(w.numel() for w in self.weights))
        lookup = None
        if self.fuse_pack and self.prepack_weights:
            lookup = self._pack_state(dev).get('by_grad', {}).get
        return adam, lookup

    def _apply_update(self, grad_scale, dev):
        """The optimizer step behind an exchange: ONE launch (scale + Adam + gradient clear + packed operands + loss tail) when
        the layers of the last backward pass cover every parameter, else the plain optimizer launch."""
        ent = getattr(self, '_wb_done', None)
        self._pack_done = False
        if ent and self.fuse_adam and self.batch_wgrad and self.optimizer is not None:
            adam, lookup = self._adam_args(dev, grad_scale)
            if ops.apply_wgrad_batch(adam, lookup, entries=ent):
                self._pack_done = bool(ops.PACK_FUSED)
                return True
        ops.finish_loss_tail()
        self.optimizer.apply(self._flat_params, self._flat_grads, grad_scale=grad_scale, zero_grads=True)
        return False

    def _loss_and_backward_gen(self, inputs, targets, train=True, fuse_update=None, split=False, skip_pack=False, exchange=False):
        """Generator form of the step: with split=True (and a usable cut, see _plan_exchange) it yields ONCE, after the
        gradients of bucket A (decoder side) are final in the flat gradient buffer, so that the caller can start their
        all-reduce -- or end a hipGraph capture -- before the encoder-side half of the backward pass is issued.  Returns the
        stats tensor (StopIteration.value)."""
        self._update_done = False
        self._did_split = False
        if train and self.batch_wgrad and inputs and inputs[0].is_cuda and not _nat.lds_oob_reads_zero(inputs[0].device):
            # the batched weight gradient masks slab tails by LDS address (wgrad_batch.hip): a device that failed the probe
            # (warned once by _native) takes the per-layer weight-gradient kernels
            self.batch_wgrad = False
        if len(targets) != len(self.outputs):
            raise ValueError('Error when checking model target: expected %d target arrays, got %d'
                             % (len(self.outputs), len(targets)))
        if self._cf_model:
            # (mse / mae do not depend on the layout: the loss is formed channels_last, the fused head's layout)
            targets = [ops.channels_first_to_last(t) if t.dim() == 5 else t for t in targets]
        fuse = None
        if train and self.fuse_head_loss:
            fuse = {o.uid: (t, w) for o, t, w in zip(self.outputs, targets, self.loss_weights)}
            ops.DIRECT_PARAM_GRADS = True       # (head_mse_applicable checks it: the fused step needs the flat gradient buffer)
        cutplan = self._plan_exchange() if (train and split) else None
        self._record_cut = cutplan[0] if cutplan else None
        # the loss's last reduction stage rides in the step's last launch when that is the fused reduction + optimizer
        ops.finish_loss_tail()
        dp_fused = bool(train and exchange and fuse_update is not None and not cutplan)
        ops.DEFER_LOSS_TAIL = bool(train and fuse_update is not None and self.fuse_adam and self.batch_wgrad and self.fold_loss_tail
                                   and self.optimizer is not None and (self._world == 1 or dp_fused) and not cutplan)
        self._pack_done = False
        self._loss_in_cl = self._cf_model       # (the predictions stay channels_last for the loss)
        try:
            # skip_pack: the packed operands are current (_ensure_packed) and this step's own last launch keeps them so
            outs = self._forward(inputs, repack=not skip_pack, fuse_targets=fuse)
        finally:
            self._loss_in_cl = False
            self._record_cut = None
            ops.DEFER_LOSS_TAIL = False
            if not train:
                ops.finish_loss_tail()
            if fuse is not None:
                ops.DIRECT_PARAM_GRADS = False
        fused = getattr(self, '_fused_outputs', set()) if fuse is not None else set()
        stats = [o if out.uid in fused else ops.mse_mae(o, t, w)
                 for out, o, t, w in zip(self.outputs, outs, targets, self.loss_weights)]
        if train:
            dev = stats[0].device
            ones = [ops.unit_seed(dev) for _ in stats]
            ops.DIRECT_PARAM_GRADS = True       # weight gradients accumulate straight into the flat gradient buffer
            ops.WGRAD_SIDE_STREAM = self.wgrad_side_stream
            ops.DEFER_WGRAD_REDUCE = self.defer_wgrad_reduce    # ... through ONE reduction launch for all layers
            # ... and where the batched kernel applies (bf16, pre-masked or activation-free layers) the weight gradients of all
            # layers are ONE launch after the last data gradient
            ops.WGRAD_BATCH = self.batch_wgrad and not self.wgrad_side_stream
            ops.drop_deferred_reduce()
            ops.drop_wgrad_batch()
            ops.drop_pending_rings()
            try:
                cut = [t for t in self._cut_tensors] if cutplan else []
                self._cut_tensors = []
                if cut:
                    # bucket A: backward pass down to the cut, its weight gradients reduced; then the caller's turn
                    gcut = torch.autograd.grad(stats, cut, ones, allow_unused=True)
                    ops.flush_wgrad_batch(None)
                    ops.flush_deferred_reduce(dev)
                    self._did_split = True
                    yield None                      # (GeneratorExit here runs the `finally` below: the switches go back)
                    pairs = [(t, g) for t, g in zip(cut, gcut) if g is not None]
                    if pairs:
                        torch.autograd.backward([t for t, _ in pairs], [g for _, g in pairs])
                else:
                    torch.autograd.backward(stats, ones)
                if dp_fused:
                    # data parallel: local reduction -> all-reduce of the flat buffer -> ONE launch that applies the update
                    ops.flush_wgrad_batch(None)
                    ops.flush_deferred_reduce(dev)
                    self._wb_done = ops.flushed_wgrad_entries()
                    parallel.allreduce_gradients(self._flat_grads)
                    self._update_done = True
                    self._apply_update(fuse_update, dev)
                else:
                    adam = lookup = None
                    if (fuse_update is not None and self.fuse_adam and not ops._deferred and self.optimizer is not None
                            and self._world == 1 and not cut):
                        adam, lookup = self._adam_args(dev, fuse_update)
                    self._update_done = ops.flush_wgrad_batch(adam, lookup)
                    self._pack_done = bool(self._update_done and ops.PACK_FUSED)
                    ops.flush_deferred_reduce(dev)
                    self._wb_done = ops.flushed_wgrad_entries()
            finally:
                ops.DIRECT_PARAM_GRADS = False
                ops.WGRAD_SIDE_STREAM = False
                ops.DEFER_WGRAD_REDUCE = False
                ops.WGRAD_BATCH = False
                ops.drop_deferred_reduce()
                ops.drop_wgrad_batch()
                if ops._pending_ring:
                    ops.drop_pending_rings()
                    if sys.exc_info()[0] is None:   # (not on top of an exception of the backward pass itself)
                        raise RuntimeError('a deferred ring fix-up was not consumed by its pooling node (plan error)')
                ops.finish_loss_tail()              # (no fused reduction + optimizer launch took it: its own launch)
                ops.join_side_stream(stats[0].device)
        if len(stats) == 1:
            return stats[0].detach().view(1, 2)                 # no copy launch for the single-output case
        return torch.stack([s.detach() for s in stats])

    def _exchange_slices(self):
        """(bucket A, bucket B) views of the flat gradient buffer (see _plan_exchange)."""
        lo = self._plan_exchange()[1]
        return self._flat_grads[lo:], self._flat_grads[:lo]

    # -- weight regularizers / constraints (CubeSphereConv2D(kernel_regularizer=..., kernel_constraint=...), DLWP/custom.py:837-842,
    #    898-914).  Off the hot path: a model that has any trains through the eager step (no captured graph, optimizer not fused into
    #    the reduction); the penalty's gradient joins the flat gradient buffer before the update, constraints follow it, and the
    #    penalty itself travels as one extra row of the step's statistics ('loss' includes it, like keras).
    def _weight_rules(self):
        rules = getattr(self, '_wrules', None)
        if rules is None:
            rules = []
            for lay in self._weight_layers():
                for w, (reg, con) in zip(lay._weights, getattr(lay, '_weight_rules', [])):
                    if w.requires_grad and (reg is not None or con is not None):
                        rules.append((w, reg, con))
            self._wrules = rules
        return rules

    def _grad_view(self, w):
        off = (w.data_ptr() - self._flat_params.data_ptr()) // 4
        return self._flat_grads[off:off + w.numel()]

    def _penalty_row(self, add_grads, grad_scale=1.0):
        """(1, 2) device tensor [sum of the regularization penalties, 0]; add_grads: their gradients join the flat buffer"""
        from .._native import check, lib, ptr, stream_ptr
        pen = torch.zeros((1, 2), dtype=torch.float32, device=self._flat_params.device)
        for w, reg, _ in self._weight_rules():
            if reg is None or (reg.l1 == 0.0 and reg.l2 == 0.0):
                continue
            g = self._grad_view(w) if add_grads else None
            check(lib().dlwpcs_l1l2_regularize(ptr(w), ptr(g), w.numel(), reg.l1, reg.l2, 1.0 / float(grad_scale), ptr(pen), stream_ptr()),
                  'dlwpcs_l1l2_regularize')
        return pen

    def _apply_constraints(self):
        for w, _, con in self._weight_rules():
            if con is not None:
                con.apply(w)

    def _update_with_rules(self, scale):
        pen = self._penalty_row(True, scale) if self._weight_rules() else None
        self.optimizer.apply(self._flat_params, self._flat_grads, grad_scale=scale)
        if pen is not None:
            self._apply_constraints()
        self._packed_ok = False
        return pen

    def _apply_gradients(self):
        scale = parallel.allreduce_gradients(self._flat_grads)  # RCCL over xGMI: one flat 2.7 MB buffer per step
        return self._update_with_rules(scale)

    def _train_step_eager(self, inputs, targets):
        self._flat_grads.zero_()
        self._grads_clean = False                               # the gradients stay readable after an eager step
        # (an eager step keeps its gradients readable: the optimizer is not fused into the reduction here)
        if self._split_wanted():
            gen = self._loss_and_backward_gen(inputs, targets, True, None, split=True)
            ha = None
            try:
                next(gen)                                       # ... the gradients of bucket A are final
                ha = parallel.allreduce_start(self._exchange_slices()[0])
                next(gen)
                raise RuntimeError('the split step yielded twice')
            except StopIteration as e:
                stats = e.value
            finally:
                gen.close()                                     # (a failure between the halves: the step's global switches go back)
            if self._did_split:
                hb = parallel.allreduce_start(self._exchange_slices()[1])
                parallel.allreduce_wait(ha)
                parallel.allreduce_wait(hb)
                pen = self._update_with_rules(1.0 / self._world)
                return stats if pen is None else torch.cat([stats, pen], dim=0)
            pen = self._apply_gradients()
            return stats if pen is None else torch.cat([stats, pen], dim=0)
        stats = self._loss_and_backward(inputs, targets, True)
        pen = self._apply_gradients()
        return stats if pen is None else torch.cat([stats, pen], dim=0)

    def train_on_device_batch(self, inputs, targets):
        """
        One optimisation step on device-resident tensors.  Static shapes are captured in a hipGraph on their second
        occurrence and replayed afterwards.  Returns the (n_out, 2) device tensor of [weighted mse, mae].
        """
        if not self._compiled:
            raise RuntimeError('You must compile your model before training/testing. Use `model.compile(...)`.')
        # (the engine options in force are part of the key: a step captured under one DLWPCS_OPTIONS string is not replayed under another)
        key = tuple(tuple(t.shape) for t in inputs + targets) + (os.environ.get('DLWPCS_OPTIONS', ''),)
        if not self.use_graphs or self._weight_rules():
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
        if g.get('no_pack'):
            self._ensure_packed(g['inputs'][0].device)          # (a launch only if something else changed the parameters)
        g['fwd_bwd'].replay()
        if g['update'] is not None:
            if g.get('bwd_b') is not None:
                # two buckets: the decoder-side gradients travel while the encoder-side half of the backward pass runs
                sa, sb = self._exchange_slices()
                ha = parallel.allreduce_start(sa)
                g['bwd_b'].replay()
                hb = parallel.allreduce_start(sb)
                parallel.allreduce_wait(ha)
                parallel.allreduce_wait(hb)
            else:
                parallel.allreduce_gradients(self._flat_grads)  # between the two graphs (scale is baked into 'update')
            g['update'].replay()
        self._grads_clean = True                                # the optimizer launch also cleared the gradient buffer
        self._packed_ok = bool(g.get('no_pack'))                # ... and, in a step without packing launch, kept the packed copies
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
        g1, g2 = _new_graph(), _new_graph()
        g1b = None
        no_pack_entry = False
        split = self._split_wanted()
        # the gradient buffer is cleared by the optimizer launch of the previous replay (DLWPCS_ADAM_ZERO_GRAD), once
        # here for the first one: the step graph needs no fill launch
        self._flat_grads.zero_()
        self._grads_clean = True
        # No cyclic garbage collection while a stream is capturing: a collector pass that happens to run inside the capture
        # would destroy whatever garbage it finds there (older models' graphs, pools, streams) through HIP calls that are
        # not allowed during capture -- seen as a rare abort() of the process.  torch.cuda.graph collects once on entry.
        gc_was_enabled = gc.isenabled()
        gc.collect()
        gc.disable()
        # With a process group alive its watchdog thread issues HIP calls of its own: only THIS thread's calls are held to the
        # capture rules then (a foreign call must not invalidate the capture).
        mode = 'thread_local' if (self._world > 1 or parallel.group_alive()) else 'global'
        try:
            if split:
                # forward + decoder-side backward | encoder-side backward: two graphs, the first bucket's all-reduce is
                # started between them (train_on_device_batch)
                gen = self._loss_and_backward_gen(static_in, static_tg, True, None, split=True)
                stats = None
                try:
                    with torch.cuda.graph(g1, capture_error_mode=mode):
                        try:
                            next(gen)
                        except StopIteration as e:
                            stats = e.value
                    if stats is None:
                        g1b = _new_graph()
                        with torch.cuda.graph(g1b, pool=g1.pool(), capture_error_mode=mode):
                            try:
                                next(gen)
                                raise RuntimeError('the split step yielded twice')
                            except StopIteration as e:
                                stats = e.value
                finally:
                    gen.close()
            else:
                # One graph per step.  World size 1: the step's last launch is the fused reduction + optimizer.  Data parallel
                # (round 4): local reduction -> all-reduce CAPTURED in the graph -> one launch that applies the update
                # (scale + Adam + gradient clear + packed operands + loss tail): 31 + 1 launches and the collective, one replay,
                # no inter-graph gap.  DLWPCS_DP_ONE_GRAPH=0 (or a capture that fails) falls back to two graphs with the
                # all-reduce issued by the host between them.
                # Either way the step's last launch can keep the packed operands current, the captured step then starts without
                # the packing launch; captured again WITH it if that launch turns out not to cover every layer.
                dp = parallel.exchange_wanted()
                one = (not dp) or (parallel.exchange_capturable() and os.environ.get('DLWPCS_DP_ONE_GRAPH', '1') == '1')
                done = False
                if one:
                    try:
                        no_pack = (self.fuse_pack and self.fuse_adam and self.batch_wgrad and self.prepack_weights
                                   and static_in[0].is_cuda)
                        if no_pack:
                            self._ensure_packed(static_in[0].device)
                            gtry = _new_graph()
                            with torch.cuda.graph(gtry, capture_error_mode=mode):
                                stats = self._loss_and_backward(static_in, static_tg, True, fuse_update=grad_scale, skip_pack=True,
                                                                exchange=dp)
                            if self._update_done and self._pack_done:
                                g1 = gtry
                                no_pack_entry = True
                            else:
                                no_pack = False
                                del gtry
                        if not no_pack:
                            with torch.cuda.graph(g1, capture_error_mode=mode):
                                stats = self._loss_and_backward(static_in, static_tg, True, fuse_update=grad_scale, exchange=dp)
                                if not self._update_done:
                                    # no exchange step: the update rides in the same graph (no inter-graph gap); normally INSIDE
                                    # the reduction of the batched weight gradients, as a launch of its own when those do not
                                    # cover every parameter
                                    self.optimizer.apply(self._flat_params, self._flat_grads, grad_scale=grad_scale, zero_grads=True)
                        done = True
                        g2 = None
                    except Exception as exc:
                        if not dp:
                            raise
                        import warnings
                        warnings.warn('capturing the gradient all-reduce inside the step graph failed (%s: %s); falling back to '
                                      'two graphs with the exchange between them' % (type(exc).__name__, exc))
                        torch.cuda.synchronize()
                        g1 = _new_graph()
                        no_pack_entry = False
                        self._flat_grads.zero_()
                if one and dp and self._world > 1 and not parallel.all_ranks_agree(done):
                    # some rank could not capture the exchange inside its step graph: EVERY rank takes the two-graph form (a rank
                    # replaying a graph with a collective inside would wait for peers that issue theirs from the host)
                    if done:
                        torch.cuda.synchronize()
                        g1, g2 = _new_graph(), _new_graph()
                        no_pack_entry = False
                        self._flat_grads.zero_()
                    done = False
                if not done:
                    with torch.cuda.graph(g1, capture_error_mode=mode):
                        stats = self._loss_and_backward(static_in, static_tg, True, fuse_update=None)
            if g2 is not None:
                dev = static_in[0].device
                if getattr(self, '_wb_done', None) and self.fuse_adam and self.batch_wgrad:
                    # (the union of the two buckets' layers: its plan has not been uploaded yet)
                    ops.prepare_wgrad_plan(self._wb_done, sum(w.numel() for w in self.weights))
                with torch.cuda.graph(g2, pool=g1.pool(), capture_error_mode=mode):
                    self._apply_update(grad_scale, dev)
        finally:
            if gc_was_enabled:
                gc.enable()
        entry = {'fwd_bwd': g1, 'bwd_b': g1b, 'update': g2, 'inputs': static_in, 'targets': static_tg, 'stats': stats,
                 'grad_scale': grad_scale, 'no_pack': no_pack_entry}
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
        if isinstance(arr, staging.LazyTake):
            arr = arr.materialise()
        a = np.ascontiguousarray(arr, dtype=np.float32)
        t = torch.from_numpy(a).to(backend.device(), non_blocking=False)
        return t if dt == torch.float32 else t.to(dt)

    def _resident_plan(self, x, y, batch_size, epochs_left, steps_per_epoch):
        """fit() on host arrays over several epochs: the batches of the FIRST epoch travel over PCIe as always (pinned staging, copy
        stream) and are kept, in the order they went up, in device buffers sized for the whole data set; every later epoch draws its
        batches out of HBM (one gather launch per tensor) -- the link is crossed once per fit() instead of once per epoch (24.8 MB up
        per bf16 / fp32 batch of config 3 takes longer than the 0.65-ms step: host-fed epochs run at half the resident rate).
        Only when the arrays, as stored (inputs in the compute dtype, targets fp32), fit into 40 % of the free HBM; None otherwise."""
        if (backend.device().type != 'cuda' or not option('host_staging') or not option('resident_data') or y is None
                or epochs_left < 2 or steps_per_epoch is not None):
            return None
        try:
            xs, ys = self._standardize_inputs(x), self._standardize_targets(y)
        except Exception:
            return None
        if not all(isinstance(a, np.ndarray) for a in xs + ys) or not xs:
            return None
        n = xs[0].shape[0]
        if any(a.shape[0] != n for a in xs + ys) or n < 2:
            return None
        cdt = backend.torch_dtype(self.compute_dtype)
        esz = 2 if cdt == torch.bfloat16 else 4
        need = sum(a.size * esz for a in xs) + sum(a.size * 4 for a in ys)
        free, _ = torch.cuda.mem_get_info(backend.device())
        if need > 0.4 * free:
            return None
        return {'state': 'fill', 'n': n, 'bufs': None, 'order': np.empty(n, dtype=np.int64), 'filled': 0, 'bytes': need}

    def _feed(self, x, y, batch_size, shuffle, resident=None):
        """(device inputs, device targets) per batch.  On a HIP device host arrays go through pinned staging buffers and a copy
        stream (keras/staging.py): the upload of a batch overlaps the training of the one before.  `resident` (_resident_plan):
        the first epoch fills device buffers with what it uploads, later epochs gather their batches from them."""
        if resident is not None and resident['state'] == 'ready':
            dev = backend.device()
            n, nx = resident['n'], len(self.inputs)
            bs = min(32, n) if batch_size is None else int(batch_size)
            idx = np.arange(n)
            if shuffle:
                np.random.shuffle(idx)              # (the draw _batches_from makes: the same batches as a host-fed epoch)
            pos = resident['inv'][idx]
            contiguous = bool(np.array_equal(pos, np.arange(n)))
            pos_dev = None if contiguous else torch.from_numpy(pos).to(dev)
            for s in range(0, n, bs):
                if contiguous:
                    ts = [b[s:s + bs] for b in resident['bufs']]
                else:
                    sel = pos_dev[s:s + bs]
                    ts = [b.index_select(0, sel) for b in resident['bufs']]
                yield ts[:nx], ts[nx:]
            return
        if backend.device().type != 'cuda' or not option('host_staging'):
            for bx, by in self._batches_from(x, y, batch_size, shuffle):
                yield [self._to_device(a) for a in bx], [self._to_device(a, target=True) for a in by]
            return
        dev = backend.device()
        if self._stager is None or self._stager.device != dev:
            self._stager = staging.Stager(dev)
        cdt = backend.torch_dtype(self.compute_dtype)
        consumed = []                   # events behind the steps that read the last batches: bounds the run-ahead of the host
        for bx, by in self._batches_from(x, y, batch_size, shuffle, lazy=True):
            if len(consumed) >= 3:
                consumed.pop(0).synchronize()
            items = list(bx) + list(by)
            ts, ev = self._stager.upload(items, [cdt] * len(bx) + [torch.float32] * len(by))
            cur = torch.cuda.current_stream(dev)
            cur.wait_event(ev)
            for t, src in zip(ts, items):
                if t is not src:                    # (device-resident items were passed through untouched)
                    t.record_stream(cur)
            # targets the caller already holds on the device in another dtype: converted here, on the compute stream
            ts = ts[:len(bx)] + [t if t.dtype == torch.float32 else t.float() for t in ts[len(bx):]]
            if resident is not None and resident['state'] == 'fill':
                sel = getattr(bx[0], 'sel', None)
                k, nb = resident['filled'], ts[0].shape[0]
                if sel is None or k + nb > resident['n']:
                    resident['state'] = 'off'           # (not the lazily gathered numpy batches this was planned for)
                else:
                    if resident['bufs'] is None:
                        resident['bufs'] = [torch.empty((resident['n'],) + tuple(t.shape[1:]), dtype=t.dtype, device=dev) for t in ts]
                    for b, t in zip(resident['bufs'], ts):
                        b[k:k + nb].copy_(t)            # (device to device, on the compute stream, in front of the step that reads t)
                    resident['order'][k:k + nb] = np.arange(sel.start, sel.stop) if isinstance(sel, slice) else np.asarray(sel)
                    resident['filled'] = k + nb
            yield ts[:len(bx)], ts[len(bx):]
            ev2 = torch.cuda.Event()
            ev2.record(torch.cuda.current_stream(dev))          # (the consumer has enqueued its step by now)
            consumed.append(ev2)
        if resident is not None and resident['state'] == 'fill':
            if resident['filled'] == resident['n'] and len(set(resident['order'].tolist())) == resident['n']:
                inv = np.empty(resident['n'], dtype=np.int64)
                inv[resident['order']] = np.arange(resident['n'])
                resident['inv'] = inv
                resident['state'] = 'ready'
            else:
                resident['state'] = 'off'
                resident['bufs'] = None

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

    def _batches_from(self, x, y, batch_size, shuffle, lazy=False):
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
        def pick(a, sel, sl):
            if lazy and isinstance(a, np.ndarray):
                return staging.LazyTake(a, sel if shuffle else sl)      # gathered straight into pinned memory later
            return a[sel] if shuffle else a[sl]
        for s in range(0, n, bs):
            sel, sl = idx[s:s + bs], slice(s, min(s + bs, n))
            yield [pick(a, sel, sl) for a in xs], [pick(a, sel, sl) for a in ys]

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
        self.release_rollout_buffers()      # (a rollout's series buffer + captured chain: up to rollout_keep_bytes of HBM)
        resident = self._resident_plan(x, y, batch_size, epochs - initial_epoch, steps_per_epoch)
        self._last_resident = resident      # (tests / benches read its state; the buffers die with this fit() call's last reference)
        cbl.call('on_train_begin', None)
        dev = backend.device()
        for epoch in range(initial_epoch, epochs):
            if self.stop_training:
                break
            t0 = time.time()
            cbl.call('on_epoch_begin', epoch, None)
            sums = torch.zeros((len(self.outputs) + (1 if self._weight_rules() else 0), 2), dtype=torch.float32, device=dev)
            count = 0
            feed = self._feed(x, y, batch_size, shuffle, resident)
            try:
                for bi, (dx, dt) in enumerate(feed):
                    if steps_per_epoch is not None and bi >= steps_per_epoch:
                        break
                    if cbl.wants_batch_logs:
                        cbl.call('on_train_batch_begin', bi, None)
                    if count == 0 and epoch == initial_epoch:
                        self._check_shapes(dx, self.inputs, 'input')
                        self._check_shapes(dt, self.outputs, 'target')
                    if self.check_finite:
                        # The fused activation evaluates ReLU(alpha, max) as min(max(x, alpha x), max): a NaN pre-activation
                        # comes out as `max`, so a NaN in the data does not necessarily reach the loss.  DLWPCS_CHECK_FINITE=1
                        # checks every batch on the device (one reduction + a host sync per step).
                        for a in list(dx) + list(dt):
                            if not bool(torch.isfinite(a).all()):
                                raise FloatingPointError('non-finite values in batch %d of epoch %d' % (bi, epoch))
                    stats = self.train_on_device_batch(dx, dt)
                    sums += stats
                    count += 1
                    if cbl.wants_batch_logs:
                        vals = self._assemble_logs(stats, 1)
                        cbl.call('on_train_batch_end', bi, dict(zip(names, vals), batch=bi, size=int(dx[0].shape[0])))
            finally:
                feed.close()
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
        if resident is not None:
            resident['bufs'] = None             # (the HBM copy of the data set lives for this call only)
        cbl.call('on_train_end', None)
        return history

    def fit_generator(self, generator, **kwargs):
        return self.fit(generator, **kwargs)

    def _evaluate_impl(self, x, y, batch_size, steps):
        if isinstance(x, tuple) and y is None and len(x) in (2, 3) and not hasattr(x, 'shape'):
            x, y = x[0], x[1]
        dev = backend.device()
        rules = bool(self._weight_rules())
        sums = torch.zeros((len(self.outputs) + (1 if rules else 0), 2), dtype=torch.float64, device=dev)
        total = 0
        feed = self._feed(x, y, batch_size, False)
        try:
            with torch.no_grad():
                for bi, (dx, dt) in enumerate(feed):
                    if steps is not None and bi >= steps:
                        break
                    n = dx[0].shape[0]
                    stats = self._loss_and_backward(dx, dt, train=False)
                    if rules:
                        stats = torch.cat([stats, self._penalty_row(False)], dim=0)
                    sums += stats.double() * n            # keras weights batches by their size
                    total += n
        finally:
            feed.close()
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
        dev = backend.device()
        staged = dev.type == 'cuda' and option('host_staging')
        if staged:
            # inputs through pinned memory + the copy stream, results back through a second copy stream: upload, forward pass and
            # download of neighbouring batches overlap (keras/staging.py)
            if self._stager is None or self._stager.device != dev:
                self._stager = staging.Stager(dev)
            if getattr(self, '_downloader', None) is None or self._downloader.device != dev:
                self._downloader = staging.Downloader(dev)       # (its pinned buffers are reused by every predict() call)
            down = self._downloader
            cdt = backend.torch_dtype(self.compute_dtype)
        with torch.no_grad():
            for s in range(0, n, bs):
                if staged:
                    srcs = [staging.LazyTake(a, slice(s, min(s + bs, n))) if isinstance(a, np.ndarray) else a[s:s + bs]
                            for a in xs]
                    dx, ev = self._stager.upload(srcs, [cdt] * len(xs))
                    cur = torch.cuda.current_stream(dev)
                    cur.wait_event(ev)
                    for t, src in zip(dx, srcs):
                        if t is not src:
                            t.record_stream(cur)
                else:
                    dx = [self._to_device(a[s:s + bs]) for a in xs]
                if s == 0:
                    self._check_shapes(dx, self.inputs, 'input')
                res = self._forward(dx, repack=(s == 0))
                if outs is None:
                    outs = [np.empty((n,) + tuple(r.shape[1:]), dtype=np.float32) for r in res]
                for o, r in zip(outs, res):
                    if staged:
                        down.push(o[s:s + bs], r)
                    else:
                        o[s:s + bs] = r.float().cpu().numpy()
            if staged:
                down.flush()
        if outs is None:
            outs = [np.empty((0,) + tuple(o.shape[1:]), dtype=np.float32) for o in self.outputs]
        return outs[0] if self._single_output else outs

    def invalidate_packed(self):
        """Tell the model that its parameters were changed behind its back.  Replayed training steps without a packing launch
        (`fuse_pack`) trust the packed bf16 operands their own optimizer launch left behind; they notice parameter changes made
        by this class (`set_weights`, `load_weights`, eager steps, other graphs) and in-place torch operations on the flat
        buffer's VIEWS that move its version counter -- NOT writes through `.data`, raw pointers, a native kernel, or another
        object's optimizer on the same buffer.  After such a write call this method: the next pass packs first."""
        self._packed_ok = False
        st = getattr(self, '_pack_cache', None)
        if st is not None:
            st['packed'] = False

    def _ensure_packed(self, device):
        """A captured step without a packing launch relies on the packed operands being current: true after such a step itself
        (its optimizer launch refreshed them), not after anything else that changed the parameters -- an optimizer launch of its
        own (eager steps, other graphs; they clear _packed_ok) or in-place torch operations on the parameter tensors (set_weights,
        load_weights: they move the flat buffer's version counter)."""
        st = self._pack_state(device)
        # (a model that was never compiled has no flat parameter buffer whose version counter could be watched: it packs every time)
        ver = self._flat_params._version if self._flat_params is not None else None
        if st['n'] and not (ver is not None and self._packed_ok and st.get('packed') and st.get('packed_version') == ver):
            ops.pack_batch(st['items'], st['n'])
            st['packed'] = True
            st['packed_version'] = ver
            self._packed_ok = True

    def predict_on_device(self, inputs, repack=True, padded_io=False):
        """Forward pass on device tensors without host round trips (used by the device-resident rollout).
        repack=False skips the weight-packing launch: only for consecutive passes with unchanged weights.
        padded_io (bf16 rollouts whose channel count is not a multiple of 8, e.g. 26 = 13 variables x 2 steps): a pointwise
        output layer writes its rows padded with zero channels to the next multiple of 8 -- and such a tensor is accepted as the
        model's input (the first convolution reads 64-B aligned rows instead of 52-B ones: 85 -> 58 us at N = 96).  Slice
        `[..., :C]` to get the reference layout."""
        if padded_io and not self._padded_io_ok:
            raise ValueError('padded_io needs a model whose single input is read by one fused convolution as its plain source')
        self._padded_io = bool(padded_io)
        try:
            with torch.no_grad():
                outs = self._forward(_as_list(inputs), repack=repack)
        finally:
            self._padded_io = False
        return outs[0] if self._single_output else outs

    def rollout_passes_on_device(self, state, passes, series=None, n_steps=1):
        """
        `passes` forward passes state -> model(state) -> ... with the state resident in HBM (the inner loop of the reference's
        predict_timeseries, DLWP/model/models.py:446-454, without its per-step numpy round trip); returns the last state.
        `series` (device, fp32, (passes * n_steps,) + state.shape): output k of pass t is written to series[t * n_steps + k].
        With use_graphs the whole chain -- passes x (the plan's launches + the series copies) -- is captured into ONE hipGraph on
        the second call with the same shapes and replayed afterwards: no launch gaps between the ~11 kernels of a pass, one
        graph launch per rollout (BASELINE config 5: 20 passes = 221 launches).  The weights must not change between the
        capture and a replay without going through this model (set_weights / an optimizer step re-pack before the replay).
        The returned state lives in the graph's memory pool: it is overwritten by the next call with the same key.
        """
        passes = int(passes)
        C = state.shape[-1]
        padded = (state.dtype == torch.bfloat16 and C % 8 != 0 and self._padded_io_ok
                  and option('padded_io'))
        shape0 = tuple(state.shape)

        def chain(x, repack_first):
            self._padded_io = padded
            try:
                for t in range(passes):
                    res = self._forward([x], repack=(repack_first and t == 0))
                    if tuple(res[-1].shape[:-1]) != shape0[:-1] or res[-1].shape[-1] not in (C, (C + 7) // 8 * 8):
                        raise ValueError('could not broadcast model output of shape %s into the input of shape %s'
                                         % (tuple(res[-1].shape), shape0))
                    x = res[-1]
                    if series is not None:
                        for k in range(n_steps):
                            series[t * n_steps + k].copy_(res[k][..., :C])
            finally:
                self._padded_io = False
            return x

        graphs_ok = (self.use_graphs and state.is_cuda and passes > 0)
        key = ('rollout', shape0, str(state.dtype), passes, n_steps, os.environ.get('DLWPCS_OPTIONS', ''),
               None if series is None else (series.data_ptr(), tuple(series.shape)))
        with torch.no_grad():
            if not graphs_ok:
                return chain(state, True)
            g = self._infer_graphs.get(key)
            if g is False:
                return chain(state, True)                           # (this chain could not be captured: eager, like round 4)
            if g is None:
                n = self._seen_batch.get(key, 0)
                self._seen_batch[key] = n + 1
                if n == 0:
                    return chain(state, True)                       # warm-up: allocations, workspace growth, packing
                self._ensure_packed(state.device)
                sin = state if self.static_batch_buffers else torch.empty_like(state).copy_(state)
                torch.cuda.synchronize()
                graph = _new_graph()
                gc_was_enabled = gc.isenabled()
                gc.collect()
                gc.disable()
                try:
                    mode = 'thread_local' if parallel.group_alive() else 'global'
                    with torch.cuda.graph(graph, capture_error_mode=mode):
                        out = chain(sin, False)
                except Exception as exc:
                    # a plan step that cannot be captured (a generic layer with a host synchronisation, a user Lambda, a table
                    # uploaded at first use): the model keeps working the way it did without graphs
                    import warnings
                    warnings.warn('capturing the rollout into a hipGraph failed (%s: %s); this rollout shape runs eagerly'
                                  % (type(exc).__name__, exc))
                    del graph
                    torch.cuda.synchronize()
                    self._infer_graphs[key] = False
                    return chain(state, True)
                finally:
                    if gc_was_enabled:
                        gc.enable()
                g = {'graph': graph, 'in': sin, 'out': out}
                self._infer_graphs[key] = g
            if g['in'].data_ptr() != state.data_ptr():
                g['in'].copy_(state, non_blocking=True)
            self._ensure_packed(state.device)                       # (a launch only if something changed the parameters)
            g['graph'].replay()
            return g['out']

    def rollout_on_device(self, predictors, steps, n_steps, out_series, verbose=0, batch_size=None):
        """
        Autoregressive rollout with the state resident in HBM (replaces the per-step numpy round trip of the reference's
        predict_timeseries loop, DLWP/model/models.py:446-454).  Fills out_series[(steps*n_steps), n, ...] in place.
        """
        if len(self.inputs) != 1:
            raise NotImplementedError('rollout_on_device serves single-input models; models with solar / constants inputs '
                                      'go through rollout_with_forcing (TimeSeriesEstimator)')
        n = predictors.shape[0]
        bs = 32 if batch_size is None else int(batch_size)
        with torch.no_grad():
            for s in range(0, n, bs):
                state = self._to_device(predictors[s:s + bs])
                self._check_shapes([state], self.inputs, 'input')
                # the whole series of this batch is written into ONE device buffer and downloaded once: no host
                # synchronisation inside the step loop
                # (batches of one shape share the series buffer -- the download below synchronises -- so that the second one on
                # replays the chain captured for it)
                skey = ((steps * n_steps,) + tuple(state.shape), str(state.device))
                series = self._rollout_series.get(skey)
                if series is None:
                    self.release_rollout_buffers()                  # (one shape at a time: a C96 series of 40 steps is 18 GB)
                    series = self._rollout_series[skey] = torch.empty(skey[0], dtype=torch.float32, device=state.device)
                if verbose > 0:
                    # (the chain is one graph replay: the steps of a batch finish together)
                    print('Prediction steps 1-%d/%d (samples %d-%d of %d)' % (steps, steps, s + 1, min(s + bs, n), n))
                self.rollout_passes_on_device(state, steps, series=series, n_steps=n_steps)
                out_series[:, s:s + bs] = series.cpu().numpy()
            if self._rollout_series and next(iter(self._rollout_series.values())).numel() * 4 > self.rollout_keep_bytes:
                self.release_rollout_buffers()                      # a large series does not outlive the call (a fit() may follow)

    def release_rollout_buffers(self):
        """Free the device buffers rollouts keep between calls: the series buffer and every captured chain that writes into it
        (its graph keeps a private activation pool alive)."""
        live = {t.data_ptr() for t in self._rollout_series.values()}
        for k in [k for k in self._infer_graphs if (k[0] == 'rollout' and k[-1] is not None and k[-1][0] in live) or k[0] == 'forcing']:
            del self._infer_graphs[k]           # (a forced rollout's graph owns its series, insolation and input buffers)
            self._seen_batch.pop(k, None)
        self._rollout_series.clear()

    def _input_roles(self):
        """(main, [solar...], constants | None) indices into self.inputs: by the names the reference scripts use
        (Azure/train_cs.py:191-194,392-396: 'main_input', 'solar_<k>', 'constants'), else by rank (solar inputs carry a
        time axis: (B, T, 6, N, N, 1))."""
        names = [t.layer.name for t in self.inputs]
        main, solar, const = 0, [], None
        for i, (t, nm) in enumerate(zip(self.inputs, names)):
            if i == 0:
                continue
            if nm.startswith('solar') or len(t.shape) == len(self.inputs[0].shape) + 1:
                solar.append(i)
            elif nm.startswith('constant') or i == len(self.inputs) - 1:
                const = i
            else:
                raise ValueError('rollout_with_forcing: cannot tell the role of input %r' % nm)
        return main, solar, const

    def rollout_with_forcing(self, predictors, sequence_steps, insolation=None, start_index=None, io_time_steps=None,
                             verbose=0):
        """
        Device-resident core of the reference's TimeSeriesEstimator.predict (DLWP/model/extensions.py:252-308): iterate a
        (multi-step) model with inputs [main_input, solar_1.., constants] on its own last output, re-injecting the known
        forcing every step.  Between steps the state never leaves HBM: the new main input is ONE `dlwpcs_state_repack`
        launch (last output + insolation as the last channel of every input time step -- the numpy concatenate / transpose /
        reshape of extensions.py:281-296), the solar inputs of the later integration steps are `dlwpcs_batch_gather` launches
        out of the HBM-resident insolation array, the constants tensor is reused.

        :param predictors: the model's initial inputs (array / list of arrays or device tensors), as the generator yields them
        :param sequence_steps: how many times the whole model is applied
        :param insolation: (T, 6, N, N) fp32 array on the data's time grid (row r = time r * dt), or None for a model
            without solar forcing (the main input is then the last output itself)
        :param start_index: (B,) int: row of `insolation` of each sample's FIRST input time step
        :param io_time_steps: input (= output) time steps folded into the channel axis; default: the time axis of the
            solar inputs, else 1
        :return: fp32 device tensor (sequence_steps, n_outputs, B, 6, N, N, C_out); download it once
        """
        sequence_steps = int(sequence_steps)
        if sequence_steps < 1:
            raise ValueError('sequence_steps must be a positive integer')
        xs = self._standardize_inputs(predictors)
        i_main, i_solar, i_const = self._input_roles()
        n_out = len(self.outputs)
        dev = backend.device()
        cdt = backend.torch_dtype(self.compute_dtype)
        with torch.no_grad():
            cur = [self._to_device(a) for a in xs]
            self._check_shapes(cur, self.inputs, 'input')
            B = cur[i_main].shape[0]
            space = tuple(cur[i_main].shape[1:-1])
            c_main, c_out = cur[i_main].shape[-1], self.outputs[-1].shape[-1]
            if io_time_steps is None:
                io_time_steps = self.inputs[i_solar[0]].shape[1] if i_solar else 1
            its = int(io_time_steps)
            if insolation is not None:
                if c_out % its or c_main != c_out + its:
                    raise ValueError('rollout_with_forcing: main input has %d channels, the last output %d: expected %d '
                                     'time steps x (variables + 1 solar channel)' % (c_main, c_out, its))
                if start_index is None:
                    raise ValueError('rollout_with_forcing: `start_index` is needed with `insolation`')
                sol = insolation if isinstance(insolation, torch.Tensor) else torch.from_numpy(
                    np.ascontiguousarray(insolation, dtype=np.float32))
                sol = sol.to(dev, dtype=torch.float32).contiguous()
                if tuple(sol.shape[1:]) != space:
                    raise ValueError('rollout_with_forcing: insolation grid %s != input grid %s' % (tuple(sol.shape[1:]), space))
                sol = sol.unsqueeze(1)                                              # (T, 1, *space): one "variable"
                start = np.asarray(start_index, dtype=np.int64).reshape(-1)
                if start.shape[0] != B:
                    raise ValueError('rollout_with_forcing: start_index needs one entry per sample')
                # every row read below, checked on the host (the gather kernels trust their indices)
                last = int(start.max()) + (sequence_steps - 1) * its * n_out + (n_out - 1) * its + its - 1
                if sequence_steps > 1 and (int(start.min()) < 0 or last >= sol.shape[0]):
                    raise IndexError('rollout_with_forcing: insolation rows up to %d are needed, the array has %d'
                                     % (last, sol.shape[0]))
                zero = torch.zeros(1, dtype=torch.int32, device=dev)
                steps_ar = np.arange(its, dtype=np.int64)
            elif c_main != c_out or i_solar:
                raise ValueError('rollout_with_forcing: without insolation the last output must have the main input\'s shape')
            if insolation is not None and len(i_solar) != n_out - 1:
                raise ValueError('rollout_with_forcing: %d solar inputs for %d outputs' % (len(i_solar), n_out))
            # every gather index of the rollout in ONE device tensor, uploaded before the first launch: rows start + (s+1)*its*n_out +
            # m*its + (0..its-1) for sequence step s, output m (extensions.py:277-287) -- the loop below then never touches the host
            idx_all = None
            if insolation is not None and sequence_steps > 1:
                rows = (start[None, None, :, None] + ((np.arange(1, sequence_steps) * its * n_out)[:, None, None, None]
                                                      + (np.arange(n_out) * its)[None, :, None, None] + steps_ar[None, None, None, :]))
                idx_all = torch.from_numpy(np.ascontiguousarray(rows.reshape(sequence_steps - 1, n_out, B * its).astype(np.int32))).to(dev)

            def chain(cur, sol, idx_all, series, repack_first, zero):
                cur = list(cur)
                for s in range(sequence_steps):
                    res = self._forward(cur, repack=(repack_first and s == 0))      # weights are fixed during a rollout
                    for k in range(n_out):
                        series[s, k].copy_(res[k])
                    if s + 1 == sequence_steps:
                        break
                    if insolation is None:
                        cur[i_main] = res[-1]
                        continue
                    nxt = []
                    for m in range(n_out):
                        buf = torch.empty((B, its) + space + (1,), dtype=cdt, device=dev)
                        ops.batch_gather(sol, idx_all[s, m], zero, buf.view((B * its,) + space + (1,)), 1, 0, 1, 0, 1, True)
                        nxt.append(buf)
                    cur[i_main] = ops.state_repack(res[-1], nxt[0], its)
                    for m, i in enumerate(i_solar):
                        cur[i] = nxt[m + 1]

            if verbose > 0:
                print('Time steps 1-%d/%d (one chain of launches: the steps of a batch finish together)' % (sequence_steps, sequence_steps))
            # With use_graphs the whole chain is ONE hipGraph from the second call with the same shapes on (as rollout_passes_on_device):
            # inputs, insolation rows and gather indices are copied into the graph's static buffers, the series it returns lives in the
            # graph's pool and is overwritten by the next call with the same key (TimeSeriesEstimator downloads it at once).
            key = ('forcing', tuple(tuple(t.shape) for t in cur), str(cur[i_main].dtype), sequence_steps,
                   None if insolation is None else tuple(sol.shape), os.environ.get('DLWPCS_OPTIONS', ''))
            g = self._infer_graphs.get(key) if (self.use_graphs and cur[i_main].is_cuda) else False
            if g is None:
                n = self._seen_batch.get(key, 0)
                self._seen_batch[key] = n + 1
                if n > 0:
                    self._ensure_packed(dev)
                    sin = [torch.empty_like(t).copy_(t) for t in cur]
                    ssol = None if insolation is None else torch.empty_like(sol).copy_(sol)
                    sidx = None if idx_all is None else torch.empty_like(idx_all).copy_(idx_all)
                    sser = torch.empty((sequence_steps, n_out, B) + space + (c_out,), dtype=torch.float32, device=dev)
                    szero = torch.zeros(1, dtype=torch.int32, device=dev)
                    torch.cuda.synchronize()
                    graph = _new_graph()
                    gc_was_enabled = gc.isenabled()
                    gc.collect()
                    gc.disable()
                    try:
                        mode = 'thread_local' if parallel.group_alive() else 'global'
                        with torch.cuda.graph(graph, capture_error_mode=mode):
                            chain(sin, ssol, sidx, sser, False, szero)
                        g = self._infer_graphs[key] = {'graph': graph, 'in': sin, 'sol': ssol, 'idx': sidx, 'series': sser, 'zero': szero}
                    except Exception as exc:
                        import warnings
                        warnings.warn('capturing the forced rollout into a hipGraph failed (%s: %s); this rollout shape runs eagerly'
                                      % (type(exc).__name__, exc))
                        del graph
                        torch.cuda.synchronize()
                        g = self._infer_graphs[key] = False
                    finally:
                        if gc_was_enabled:
                            gc.enable()
            if g:
                for dst, src in zip(g['in'], cur):
                    dst.copy_(src, non_blocking=True)
                if g['sol'] is not None:
                    g['sol'].copy_(sol, non_blocking=True)
                if g['idx'] is not None:
                    g['idx'].copy_(idx_all, non_blocking=True)
                self._ensure_packed(dev)                                        # (a launch only if something changed the parameters)
                g['graph'].replay()
                return g['series']
            series = torch.empty((sequence_steps, n_out, B) + space + (c_out,), dtype=torch.float32, device=dev)
            chain(cur, sol if insolation is not None else None, idx_all, series, True, zero if insolation is not None else None)
        return series

    def reset_states(self):
        pass

    # -------------------------------------------------------------------------------------------------------------- #
    # persistence: native npz containers (no pickle); HDF5 files written by keras / h5py are READ by DLWP.keras.hdf5_lite
    # -------------------------------------------------------------------------------------------------------------- #
    @staticmethod
    def _wants_hdf5(filepath, save_format):
        """keras' rule (TF 2.1 saving_utils): save_format 'h5' / 'hdf5' / 'keras', or no format and a file name ending in
        .h5 / .hdf5 / .keras; 'npz' / 'native' force the engine's container."""
        if save_format is not None:
            fmt = str(save_format).lower()
            if fmt in ('h5', 'hdf5', 'keras'):
                return True
            if fmt in ('npz', 'native', 'dlwpcs'):
                return False
            raise ValueError("save_format %r: the engine writes 'h5' (Keras HDF5 layout) or 'npz' (native container)" % (save_format,))
        return str(filepath).lower().endswith(('.h5', '.hdf5', '.keras'))

    def _keras_layer_weights(self):
        """[(layer name, [(weight name, array)])] of ALL layers, what keras' save_weights_to_hdf5_group iterates"""
        return [(l.name, list(zip(l._weight_names, l.get_weights())) if l._weights else []) for l in self.layers]

    def save_weights(self, filepath, overwrite=True, save_format=None):
        """`save_format='h5'` (reference DLWP/custom.py:186) or a name ending in .h5 / .hdf5 / .keras: an HDF5 file in Keras'
        layout (DLWP.keras.hdf5_lite.write_keras_file; opens with h5py / `keras.Model.load_weights`).  Otherwise the native
        weights file: an uncompressed numpy `.npz` container holding one array per weight plus their keras names, loaded with
        allow_pickle=False.  load_weights reads both."""
        if not overwrite and os.path.exists(filepath):
            return
        from . import serialization
        if self._wants_hdf5(filepath, save_format):
            from . import hdf5_lite
            hdf5_lite.write_keras_file(filepath, self._keras_layer_weights())
            return
        names = [n for l in self._weight_layers() for n in l._weight_names]
        layers = [[l.name, len(l._weights)] for l in self._weight_layers()]
        serialization.save_container(filepath, self.get_weights(), {'format': 'dlwpcs-weights-2', 'names': names,
                                                                    'layers': layers})

    def _set_weights_by_layer(self, file_layers, by_name, what):
        """file_layers: ordered [(layer name, [arrays])] of the layers that own weights.  keras semantics
        (`load_weights_from_hdf5_group[_by_name]`): topological order by default, layer names with by_name=True."""
        mine = self._weight_layers()
        file_layers = [(n, ws) for n, ws in file_layers if len(ws)]
        if by_name:
            table = dict(file_layers)
            for l in mine:
                if l.name in table:
                    l.set_weights(table[l.name])
            return
        if len(file_layers) != len(mine):
            raise ValueError('You are trying to load a weight file containing %d layers into a model with %d layers.'
                             % (len(file_layers), len(mine)))
        for l, (n, ws) in zip(mine, file_layers):
            if len(ws) != len(l._weights):
                raise ValueError('Layer #%s (named "%s") expects %d weight(s), but the saved weights (%s, layer "%s") have '
                                 '%d element(s).' % (mine.index(l), l.name, len(l._weights), what, n, len(ws)))
            l.set_weights(ws)

    def load_weights(self, filepath, by_name=False):
        """Native `.npz` container, or an HDF5 weights / model file written by Keras + h5py (the reference's
        `model.save_weights(path, save_format='h5')`, DLWP/custom.py:184-191): read by the pure-Python HDF5 subset reader
        DLWP.keras.hdf5_lite -- no h5py / TensorFlow needed."""
        from . import hdf5_lite, serialization
        if hdf5_lite.is_hdf5(filepath):
            layers, _ = hdf5_lite.read_keras_weights(filepath)
            self._set_weights_by_layer([(n, [a for _, a in ws]) for n, ws in layers], by_name, 'HDF5')
            return
        arrays, meta = serialization.load_container(filepath)
        if meta.get('format') != 'dlwpcs-weights-2':
            raise ValueError('%s is not a dlwpcs weights file' % filepath)
        if by_name and meta.get('layers'):
            k, fl = 0, []
            for name, n in meta['layers']:
                fl.append((name, arrays[k:k + n]))
                k += n
            self._set_weights_by_layer(fl, True, 'npz')
            return
        self.set_weights(arrays)

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

    def _save_hdf5(self, filepath, include_optimizer):
        """keras' save_model_to_hdf5: model_config / training_config as root attributes, /model_weights, /optimizer_weights
        (Adam: iterations, then every weight's m, then every weight's v -- the order of OptimizerV2.weights)"""
        from . import hdf5_lite
        tc = opt_w = None
        if self._compiled:
            ocfg = dict(self.optimizer.get_config())
            tc = json.dumps({'optimizer_config': {'class_name': 'Adam', 'config': ocfg}, 'loss': self.loss,
                             'metrics': list(self.metrics), 'weighted_metrics': None, 'sample_weight_mode': None,
                             'loss_weights': self.loss_weights})
            st = self.optimizer.state_dict() if include_optimizer else None
            if st is not None:
                base = self._flat_params.data_ptr()
                ms, vs = [], []
                for l in self._weight_layers():
                    for w, wn in zip(l._weights, l._weight_names):
                        o = (w.data_ptr() - base) // 4
                        ms.append(('Adam/%s/m:0' % wn[:-2], st['m'][o:o + w.numel()].reshape(tuple(w.shape))))
                        vs.append(('Adam/%s/v:0' % wn[:-2], st['v'][o:o + w.numel()].reshape(tuple(w.shape))))
                opt_w = [('Adam/iter:0', np.asarray(st['step'], dtype=np.int64))] + ms + vs
        hdf5_lite.write_keras_file(filepath, self._keras_layer_weights(), model_config=json.dumps(self.to_keras_config()),
                                   training_config=tc, optimizer_weights=opt_w,
                                   extra_attrs=[('dlwpcs_compute_dtype', self.compute_dtype)])

    def save(self, filepath, overwrite=True, include_optimizer=True, save_format=None, **kwargs):
        """Model file.  A name ending in .keras / .h5 / .hdf5 (the `<name>.keras` of DLWP.util.save_model, reference
        util.py:139) or save_format='h5': an HDF5 file in Keras' layout -- graph as `model_config`, weights, compile arguments
        as `training_config`, Adam state as `optimizer_weights` -- that `keras.models.load_model` (with DLWP.custom as
        custom_objects) and this engine's load_model both read.  save_format='npz' (or any other file name): the engine's
        native npz container with the same content."""
        if not overwrite and os.path.exists(filepath):
            return
        from . import serialization
        if self._wants_hdf5(filepath, save_format):
            self._save_hdf5(filepath, include_optimizer)
            return
        meta = {'format': 'dlwpcs-model-2', 'config': self.get_config(), 'keras_config': self.to_keras_config(),
                'compile': None, 'compute_dtype': self.compute_dtype, 'n_weights': len(self.weights)}
        arrays = self.get_weights()
        if self._compiled:
            meta['compile'] = {'loss': self.loss, 'loss_weights': self.loss_weights, 'metrics': self.metrics,
                               'optimizer': self.optimizer.get_config(), 'optimizer_state': None}
            st = self.optimizer.state_dict() if include_optimizer else None
            if st is not None:
                meta['compile']['optimizer_state'] = {'step': st['step']}
                arrays = arrays + [st['m'], st['v']]
        serialization.save_container(filepath, arrays, meta)

    # -- keras functional config (what `model.to_json()` / the `model_config` attribute of a keras HDF5 file hold) ------- #
    def to_keras_config(self):
        node_of = {}                     # tensor uid -> (layer name, node index)
        calls = {}
        layers = []
        entry = {}
        for t in self.inputs + [n for n in self._nodes if n.uid not in {i.uid for i in self.inputs}]:
            lay = t.layer
            if id(lay) not in entry:
                cfg = lay.get_config()
                e = {'name': lay.name, 'class_name': type(lay).__name__, 'config': cfg, 'inbound_nodes': []}
                entry[id(lay)] = e
                layers.append(e)
            e = entry[id(lay)]
            k = calls.get(id(lay), 0)
            calls[id(lay)] = k + 1
            node_of[t.uid] = (lay.name, k)
            if t.node_inputs:
                e['inbound_nodes'].append([[node_of[i.uid][0], node_of[i.uid][1], 0, {}] for i in t.node_inputs])
        return {'class_name': 'Model',
                'config': {'name': self.name, 'layers': layers,
                           'input_layers': [[node_of[t.uid][0], node_of[t.uid][1], 0] for t in self.inputs],
                           'output_layers': [[node_of[t.uid][0], node_of[t.uid][1], 0] for t in self.outputs]},
                'keras_version': '2.2.4-tf', 'backend': 'dlwpcs'}

    @classmethod
    def from_keras_config(cls, config, custom_objects=None):
        """Rebuild the graph from a keras functional-model config (`json.loads(model.to_json())` of TF-keras 2.x, or its
        'config' member): layers by class name from DLWP.keras.layers / DLWP.custom / custom_objects, nodes in dependency
        order (the algorithm of keras' `Network.from_config`)."""
        from .. import custom
        from . import layers as klayers
        if 'config' in config and 'layers' not in config:
            config = config['config']
        table = {}
        for mod in (klayers, custom):
            for k in dir(mod):
                v = getattr(mod, k)
                if isinstance(v, type) and issubclass(v, Layer):
                    table[k] = v
        table.update(custom_objects or {})
        objs, pending = {}, []
        for lc in config['layers']:
            cname = lc['class_name']
            if cname not in table:
                raise ValueError('Unknown layer: %s' % cname)
            cfg = dict(lc['config'])
            lay = table[cname].from_config(cfg)
            objs[lc['name']] = lay
            for ni, node in enumerate(lc.get('inbound_nodes', [])):
                pending.append((lc['name'], ni, node))
        tensors = {}
        for lc in config['layers']:
            lay = objs[lc['name']]
            if isinstance(lay, InputLayer):
                tensors[(lc['name'], 0)] = KTensor(lay.batch_input_shape, layer=lay, node_inputs=(), name=lay.name)
        counts = {}
        while pending:
            progressed = False
            rest = []
            for name, ni, node in pending:
                refs = [(r[0], r[1]) for r in node]
                # a layer's nodes must be created in order (node index = call count)
                if all(r in tensors for r in refs) and counts.get(name, 0) == ni:
                    lay = objs[name]
                    ins = [tensors[r] for r in refs]
                    tensors[(name, ni)] = lay(ins if (isinstance(lay, Concatenate) or len(ins) > 1) else ins[0])
                    counts[name] = ni + 1
                    progressed = True
                else:
                    rest.append((name, ni, node))
            if not progressed:
                raise ValueError('keras config: cannot resolve the inbound nodes of %s' % sorted({p[0] for p in rest}))
            pending = rest
        ins = [tensors[(r[0], r[1])] for r in config['input_layers']]
        outs = [tensors[(r[0], r[1])] for r in config['output_layers']]
        return cls(inputs=ins[0] if len(ins) == 1 else ins, outputs=outs[0] if len(outs) == 1 else outs,
                   name=config.get('name'))

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


def _compile_from_keras_training_config(model, tc):
    opt = tc.get('optimizer_config') or {}
    ocfg = dict(opt.get('config', {}))
    if opt.get('class_name', 'Adam').lower() != 'adam':
        raise NotImplementedError('optimizer %r: the DLWP-CS engine provides Adam' % opt.get('class_name'))
    ocfg = {k: ocfg[k] for k in ('learning_rate', 'lr', 'beta_1', 'beta_2', 'epsilon', 'amsgrad', 'decay') if k in ocfg}
    metrics = tc.get('metrics') or []
    flat = []
    for m in (metrics if isinstance(metrics, (list, tuple)) else [metrics]):
        flat += list(m) if isinstance(m, (list, tuple)) else [m]
    metrics = ['mae' if m in ('mae', 'mean_absolute_error') else m for m in flat]
    model.compile(optimizer=optimizers.Adam(**ocfg), loss=tc.get('loss'), loss_weights=tc.get('loss_weights'),
                  metrics=sorted(set(metrics)) or None)


def _restore_adam_from_keras(model, group):
    """/optimizer_weights of a keras model file: [iterations] + one m per weight + one v per weight, in the order of the model's
    weights (OptimizerV2.weights of Adam; files without amsgrad).  Anything else is left alone (fresh optimizer state)."""
    from . import hdf5_lite
    names = hdf5_lite._attr_list(group, 'weight_names')
    ws = [w for l in model._weight_layers() for w in l._weights]
    if len(names) != 1 + 2 * len(ws):
        return
    arrs = [np.asarray(group[n].read()) for n in names]
    base = model._flat_params.data_ptr()
    m = np.zeros(model._flat_params.numel(), dtype=np.float32)
    v = np.zeros_like(m)
    for k, w in enumerate(ws):
        a, b = arrs[1 + k], arrs[1 + len(ws) + k]
        if tuple(a.shape) != tuple(w.shape) or tuple(b.shape) != tuple(w.shape):
            return
        o = (w.data_ptr() - base) // 4
        m[o:o + w.numel()] = a.ravel()
        v[o:o + w.numel()] = b.ravel()
    model.optimizer.load_state_dict({'m': m, 'v': v, 'step': int(np.asarray(arrs[0]).ravel()[0])}, model._flat_params)


def load_model(filepath, custom_objects=None, compile=True):
    """
    Load a model file: the engine's npz container (Model.save), or an HDF5 file written by keras' `model.save()` under
    TensorFlow (the `<name>.keras` of the reference's DLWP.util.save_model, util.py:139): graph from its `model_config`
    attribute, weights from its `model_weights` group, optimizer / loss from `training_config` -- read without h5py.
    """
    from . import hdf5_lite, serialization
    if hdf5_lite.is_hdf5(filepath):
        layers, cfg = hdf5_lite.read_keras_weights(filepath)
        if cfg is None:
            raise ValueError('%s holds weights only (no model_config): build the model and use load_weights' % filepath)
        model = Model.from_keras_config(json.loads(cfg), custom_objects=custom_objects)
        model._set_weights_by_layer([(n, [a for _, a in ws]) for n, ws in layers], False, 'HDF5')
        f = hdf5_lite.File(filepath)
        cd = f.attrs.get('dlwpcs_compute_dtype')
        if cd is not None:
            model.compute_dtype = hdf5_lite._as_str(cd)                       # (files written by this engine)
        tc = f.attrs.get('training_config')
        if compile and tc is not None:
            _compile_from_keras_training_config(model, json.loads(hdf5_lite._as_str(tc)))
            if 'optimizer_weights' in f._links:
                _restore_adam_from_keras(model, f['optimizer_weights'])
        return model
    arrays, meta = serialization.load_container(filepath)
    if meta.get('format') != 'dlwpcs-model-2':
        raise ValueError('%s is not a dlwpcs model file' % filepath)
    model = Model.from_config(meta['config'], custom_objects=custom_objects)
    model.compute_dtype = meta.get('compute_dtype', model.compute_dtype)      # the mixed-precision mode travels with it
    nw = int(meta['n_weights'])
    model.set_weights(arrays[:nw])
    cmp = meta.get('compile')
    if compile and cmp:
        model.compile(optimizer=optimizers.get(cmp['optimizer']), loss=cmp['loss'], loss_weights=cmp['loss_weights'],
                      metrics=cmp['metrics'])
        st = cmp.get('optimizer_state')
        if st is not None and len(arrays) >= nw + 2:
            model.optimizer.load_state_dict({'m': arrays[nw], 'v': arrays[nw + 1], 'step': st['step']}, model._flat_params)
    return model


def clone_model(model):
    new = Model.from_config(model.get_config())
    new.compute_dtype = model.compute_dtype
    new.set_weights(model.get_weights())
    return new
