"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU restatement (numpy + torch-CPU) of the DLWP-CS cubed-sphere hot path.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import this module; the product
(`dlwp-cs_amd/`) never does, and fails loudly when its HIP library is missing.

Parity status: the reference (/root/reference) ships no tests, golden vectors or fixtures ("parity unpinned" by
the reference itself).  This oracle is therefore pinned against outputs of the reference's
own `CubeSpherePadding2D.call` / `CubeSphereConv2D.call` bodies executed in the build container under a numpy/torch
stub of the TensorFlow symbols they use (`tests/golden/gen_golden.py` -> `tests/golden/*.npz`); the index semantics
(halo gather table, weight groups, north-pole flip) are integer-exact against those vectors.  The arithmetic of
`K.conv2d` lives in TensorFlow 2.1 (tensorflow==2.1.0, reference `environment.yml:180`, not vendored, not
installable here); it is restated as plain cross-correlation (no kernel flip, HWIO kernel, 'valid' = no padding,
'same' = TF asymmetric zero padding) with torch-CPU `conv2d`.

Every function cites the reference file:line (relative to /root/reference) whose behaviour it follows.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------------------------- #
# Halo gather table: restates CubeSpherePadding2D.call, DLWP/custom.py:1082-1308
# ---------------------------------------------------------------------------------------------------------------- #

def _rows_source(f, a, b, N, p, top):
    """
    Pass 1 of the padding layer (DLWP/custom.py:1201-1251 channels_last, :1089-1139 channels_first): the source
    (face, row, col) of halo row `a` (0..p-1, counted downwards inside the halo strip), column `b` (0..N-1) of
    face `f`, for the top (`top=True`) or bottom strip.
    """
    if f == 0:      # custom.py:1203-1209  top <- last p rows of face 4; bottom <- first p rows of face 5
        return (4, N - p + a, b) if top else (5, a, b)
    if f == 1:      # custom.py:1211-1217  transposed strips of the right-hand columns of the polar faces
        return (4, N - 1 - b, N - p + a) if top else (5, b, N - 1 - a)
    if f == 2:      # custom.py:1219-1225  doubly reversed strips
        return (4, p - 1 - a, N - 1 - b) if top else (5, N - 1 - a, N - 1 - b)
    if f == 3:      # custom.py:1227-1233  transposed strips of the left-hand columns
        return (4, b, p - 1 - a) if top else (5, N - 1 - b, a)
    if f == 4:      # custom.py:1235-1241  south pole: top <- face 2 (reversed), bottom <- face 0
        return (2, p - 1 - a, N - 1 - b) if top else (0, a, b)
    if f == 5:      # custom.py:1243-1249  north pole: top <- face 0, bottom <- face 2 (reversed)
        return (0, N - p + a, b) if top else (2, N - 1 - a, N - 1 - b)
    raise ValueError(f)


_TABLE_CACHE = {}


def halo_table(N, p):
    key = (int(N), int(p))
    if key not in _TABLE_CACHE:
        _TABLE_CACHE[key] = _halo_table_uncached(N, p)
    return _TABLE_CACHE[key].copy()


def _halo_table_uncached(N, p):
    """
    int32 array T of shape (6, M, M), M = N + 2p, such that for the reference layer
    `out[b, f, i, j, c] == in[b].reshape(6*N*N, C)[T[f, i, j], c]` (channels_last), identically for channels_first.
    Built by composing the two passes of DLWP/custom.py:1198-1308 on index triples instead of data.
    """
    M = N + 2 * p
    # pass 1: row-padded faces, shape (6, M, N) of flat source indices
    out1 = np.empty((6, M, N), dtype=np.int64)
    for f in range(6):
        for b in range(N):
            for a in range(p):
                sf, si, sj = _rows_source(f, a, b, N, p, True)
                out1[f, a, b] = (sf * N + si) * N + sj
                sf, si, sj = _rows_source(f, a, b, N, p, False)
                out1[f, N + p + a, b] = (sf * N + si) * N + sj
            for i in range(N):
                out1[f, p + i, b] = (f * N + i) * N + b
    # pass 2: columns.  Equatorial faces first (custom.py:1256-1287): periodic neighbours' row-padded edges.
    T = np.empty((6, M, M), dtype=np.int64)
    for f in range(4):
        left, right = (f - 1) % 4, (f + 1) % 4
        T[f, :, p:p + N] = out1[f]
        for a in range(p):
            T[f, :, a] = out1[left, :, N - p + a]
            T[f, :, N + p + a] = out1[right, :, a]
    # polar faces (custom.py:1289-1303): strips of the FULLY padded equatorial faces 3 and 1
    for r in range(M):
        for a in range(p):
            T[4, r, a] = T[3, 2 * p - 1 - a, r]              # custom.py:1291
            T[4, r, N + p + a] = T[1, p + a, M - 1 - r]      # custom.py:1293
            T[5, r, a] = T[3, N + a, M - 1 - r]              # custom.py:1299
            T[5, r, N + p + a] = T[1, N + p - 1 - a, r]      # custom.py:1301
    T[4, :, p:p + N] = out1[4]
    T[5, :, p:p + N] = out1[5]
    return T.astype(np.int32)


def cs_pad(x, p, data_format='channels_last'):
    """CubeSpherePadding2D.call (DLWP/custom.py:1082-1308) as one gather.  x: torch tensor or ndarray."""
    is_np = isinstance(x, np.ndarray)
    xt = torch.as_tensor(x)
    if data_format == 'channels_first':   # (B, C, 6, N, N)
        B, C, Fc, N, _ = xt.shape
        T = torch.as_tensor(halo_table(N, p).astype(np.int64)).reshape(-1)
        out = torch.index_select(xt.reshape(B, C, 6 * N * N), 2, T).reshape(B, C, 6, N + 2 * p, N + 2 * p)
    else:                                 # (B, 6, N, N, C)
        B, Fc, N, _, C = xt.shape
        T = torch.as_tensor(halo_table(N, p).astype(np.int64)).reshape(-1)
        out = torch.index_select(xt.reshape(B, 6 * N * N, C), 1, T).reshape(B, 6, N + 2 * p, N + 2 * p, C)
    assert Fc == 6
    return out.numpy() if is_np else out


# ---------------------------------------------------------------------------------------------------------------- #
# CubeSphereConv2D.call, DLWP/custom.py:921-1002  (reference-structured: six per-face conv2d calls)
# ---------------------------------------------------------------------------------------------------------------- #

def _same_pads(n, k, s, d):
    """TF 'SAME' padding for one axis: total = max((ceil(n/s)-1)*s + (k-1)*d + 1 - n, 0); extra goes after."""
    out = -(-n // s)
    total = max((out - 1) * s + (k - 1) * d + 1 - n, 0)
    return total // 2, total - total // 2


def conv2d_tf(x, kernel, strides=(1, 1), padding='valid', dilation=(1, 1)):
    """
    K.conv2d semantics (call sites DLWP/custom.py:928-935,947-954,968-975,979-986): cross-correlation, x is
    (B, H, W, C) [channels_last], kernel is HWIO (kh, kw, C_in, C_out).  Returns (B, H', W', C_out).
    """
    xc = x.permute(0, 3, 1, 2)
    w = kernel.permute(3, 2, 0, 1).contiguous()      # (torch-CPU's slow_conv2d backward needs a contiguous weight)
    if padding == 'same':
        pt, pb = _same_pads(xc.shape[2], w.shape[2], strides[0], dilation[0])
        pl, pr = _same_pads(xc.shape[3], w.shape[3], strides[1], dilation[1])
        xc = F.pad(xc, (pl, pr, pt, pb))
    elif padding != 'valid':
        raise ValueError(padding)
    y = F.conv2d(xc, w, None, stride=tuple(strides), dilation=tuple(dilation))
    return y.permute(0, 2, 3, 1)


def cs_conv2d(x, equatorial_kernel, polar_kernel, north_pole_kernel=None,
              equatorial_bias=None, polar_bias=None, north_pole_bias=None,
              strides=(1, 1), padding='valid', dilation=(1, 1), data_format='channels_last',
              flip_north_pole=True, independent_north_pole=False):
    """
    CubeSphereConv2D.call (DLWP/custom.py:921-1002).  Equatorial kernel on faces 0-3 (:926-943), polar kernel on
    face 4 (:946-962), face 5 uses the polar (or independent north-pole) kernel, wrapped in a flip of the height
    axis before and after when flip_north_pole (:965-996).  torch tensors in, torch tensor out.
    """
    if data_format == 'channels_first':
        x = x.permute(0, 2, 3, 4, 1)
    use_bias = equatorial_bias is not None
    outs = []
    for f in range(4):
        y = conv2d_tf(x[:, f], equatorial_kernel, strides, padding, dilation)
        if use_bias:
            y = y + equatorial_bias
        outs.append(y)
    y = conv2d_tf(x[:, 4], polar_kernel, strides, padding, dilation)
    if use_bias:
        y = y + polar_bias
    outs.append(y)
    k5 = north_pole_kernel if independent_north_pole else polar_kernel
    b5 = north_pole_bias if independent_north_pole else polar_bias
    x5 = torch.flip(x[:, 5], dims=(1,)) if flip_north_pole else x[:, 5]
    y = conv2d_tf(x5, k5, strides, padding, dilation)
    if use_bias:
        y = y + b5
    if flip_north_pole:
        y = torch.flip(y, dims=(1,))
    outs.append(y)
    out = torch.stack(outs, dim=1)
    if data_format == 'channels_first':
        out = out.permute(0, 4, 1, 2, 3)
    return out


# ---------------------------------------------------------------------------------------------------------------- #
# Keras stock ops used between the custom layers in the U-Net (Azure/train_cs.py:196-199,277-305)
# ---------------------------------------------------------------------------------------------------------------- #

def relu_leaky_clip(x, negative_slope=0.1, max_value=10.0):
    """keras ReLU(negative_slope, max_value) (Azure/train_cs.py:199): min(x, max) for x >= 0, slope*x for x < 0."""
    pos = torch.clamp(x, min=0.0)
    if max_value is not None:
        pos = torch.clamp(pos, max=max_value)
    return pos - negative_slope * torch.clamp(-x, min=0.0)


def avgpool_122(x):
    """AveragePooling3D((1,2,2), channels_last) (Azure/train_cs.py:197): per-face non-overlapping 2x2 mean."""
    B, Fc, H, W, C = x.shape
    return x.reshape(B, Fc, H // 2, 2, W // 2, 2, C).mean(dim=(3, 5))


def upsample_122(x):
    """UpSampling3D((1,2,2), channels_last) (Azure/train_cs.py:198): per-face nearest-neighbour x2."""
    return x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)


def glorot_uniform(rng, shape):
    """keras glorot_uniform (default kernel_initializer, DLWP/custom.py:835) for an HWIO kernel."""
    kh, kw, cin, cout = shape
    limit = math.sqrt(6.0 / (kh * kw * cin + kh * kw * cout))
    return rng.uniform(-limit, limit, size=shape)


# ---------------------------------------------------------------------------------------------------------------- #
# U-Net `unet2` (Azure/train_cs.py:277-305) with the layer table at :209-228 (skip_connections=True)
# ---------------------------------------------------------------------------------------------------------------- #

# (name, C_out multiplier of base, kernel)   order = forward order of unet2
UNET2_LAYERS = ['conv_2d_1', 'conv_2d_1_2', 'conv_2d_2', 'conv_2d_2_2', 'conv_2d_5_2', 'conv_2d_5',
                'conv_2d_6_2', 'conv_2d_6', 'conv_2d_7', 'conv_2d_7_2', 'conv_2d_8']


def unet2_channel_plan(c_in, c_out, base=32):
    """(C_in, C_out, k) of every CubeSphereConv2D in unet2, forward order (Azure/train_cs.py:209-228,277-305)."""
    b = base
    return [
        (c_in, b, 3), (b, b, 3),                 # @N     conv_2d_1, conv_2d_1_2
        (b, 2 * b, 3), (2 * b, 2 * b, 3),        # @N/2   conv_2d_2, conv_2d_2_2
        (2 * b, 4 * b, 3), (4 * b, 2 * b, 3),    # @N/4   conv_2d_5_2, conv_2d_5
        (4 * b, 2 * b, 3), (2 * b, b, 3),        # @N/2   conv_2d_6_2 (after concat), conv_2d_6
        (2 * b, b, 3), (b, b, 3),                # @N     conv_2d_7 (after concat), conv_2d_7_2
        (b, c_out, 1),                           # 1x1    conv_2d_8 'output', linear
    ]


def make_unet2_params(c_in, c_out, base=32, seed=1, dtype=torch.float64, bias_std=0.1):
    """
    Parameter list in Keras get_weights() order per layer (DLWP/custom.py:882-914): equatorial_kernel, polar_kernel,
    equatorial_bias, polar_bias.  Kernels glorot-uniform; biases N(0, bias_std) (non-zero so the bias path is tested).
    """
    rng = np.random.default_rng(seed)
    params = []
    for (ci, co, k) in unet2_channel_plan(c_in, c_out, base):
        params.append(dict(
            equatorial_kernel=torch.tensor(glorot_uniform(rng, (k, k, ci, co)), dtype=dtype),
            polar_kernel=torch.tensor(glorot_uniform(rng, (k, k, ci, co)), dtype=dtype),
            equatorial_bias=torch.tensor(rng.normal(0, bias_std, size=(co,)), dtype=dtype),
            polar_bias=torch.tensor(rng.normal(0, bias_std, size=(co,)), dtype=dtype),
        ))
    return params


def _conv_block(x, prm, pad=True):
    if pad:
        x = cs_pad(x, 1, 'channels_last')
    return cs_conv2d(x, prm['equatorial_kernel'], prm['polar_kernel'], None,
                     prm['equatorial_bias'], prm['polar_bias'], None,
                     data_format='channels_last', flip_north_pole=True, independent_north_pole=False)


def unet2_forward(x, params):
    """unet2 (Azure/train_cs.py:277-305), channels_last (B, 6, N, N, C)."""
    r = relu_leaky_clip
    x0 = r(_conv_block(x, params[0]))
    x0 = r(_conv_block(x0, params[1]))
    x1 = avgpool_122(x0)
    x1 = r(_conv_block(x1, params[2]))
    x1 = r(_conv_block(x1, params[3]))
    x2 = avgpool_122(x1)
    x2 = r(_conv_block(x2, params[4]))
    x2 = r(_conv_block(x2, params[5]))
    x2 = upsample_122(x2)
    xx = torch.cat([x2, x1], dim=-1)
    xx = r(_conv_block(xx, params[6]))
    xx = r(_conv_block(xx, params[7]))
    xx = upsample_122(xx)
    xx = torch.cat([xx, x0], dim=-1)
    xx = r(_conv_block(xx, params[8]))
    xx = r(_conv_block(xx, params[9]))
    return _conv_block(xx, params[10], pad=False)


def encoder6_forward(x, params):
    """BASELINE cfg 2: the first six convolutions of unet2 (Azure/train_cs.py:278-291)."""
    r = relu_leaky_clip
    x0 = r(_conv_block(x, params[0]))
    x0 = r(_conv_block(x0, params[1]))
    x1 = avgpool_122(x0)
    x1 = r(_conv_block(x1, params[2]))
    x1 = r(_conv_block(x1, params[3]))
    x2 = avgpool_122(x1)
    x2 = r(_conv_block(x2, params[4]))
    x2 = r(_conv_block(x2, params[5]))
    return x2


def mse_loss(y, t):
    """keras 'mse' (Azure/train_cs.py:424): mean over every element of the output."""
    return ((y - t) ** 2).mean()


def adam_step(p, g, m, v, t, lr=1e-3, b1=0.9, b2=0.999, eps=1e-7):
    """
    TF2.1 keras Adam (Azure/train_cs.py:429; defaults lr 1e-3, beta 0.9/0.999, epsilon 1e-7, no amsgrad):
    lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMA; p -= lr_t*m/(sqrt(v)+eps).  In place on torch tensors; t is 1-based.
    """
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    p.addcdiv_(m, v.sqrt().add_(eps), value=-lr_t)


# ---------------------------------------------------------------------------------------------------------------- #
# DLWPFunctional.predict_timeseries bookkeeping, DLWP/model/models.py:418-460
# ---------------------------------------------------------------------------------------------------------------- #

def predict_timeseries_ref(predict_fn, predictors, time_steps, n_steps=1, time_dim=1, keep_time_dim=False,
                           is_recurrent=False):
    """Array bookkeeping of DLWPFunctional.predict_timeseries (DLWP/model/models.py:431-460) around `predict_fn`."""
    if isinstance(predictors, (list, tuple)):
        raise NotImplementedError
    time_steps = int(time_steps)
    if time_steps < 1:
        raise ValueError("time_steps must be an int > 0")
    steps = int(np.ceil(time_steps / n_steps / time_dim))
    out_steps = steps * n_steps
    series = np.full((out_steps,) + predictors.shape, np.nan, dtype=np.float32)
    p = predictors.copy()
    sample_dim = p.shape[0]
    feature_shape = p.shape[2:] if is_recurrent else p.shape[1:]
    for t in range(steps):
        result = predict_fn(p)
        if n_steps == 1:
            p[:] = result[:]
        else:
            p[:] = result[-1]
        series[t * n_steps:(t + 1) * n_steps, ...] = np.stack(result, axis=0) if n_steps > 1 else result
    series = series.reshape((out_steps, sample_dim, time_dim, -1) + feature_shape[1:])
    if not keep_time_dim:
        series = series.transpose((0, 2, 1) + tuple(range(3, 3 + len(feature_shape))))
        series = series.reshape((out_steps * time_dim, sample_dim, -1) + feature_shape[1:])
    return series


# ---------------------------------------------------------------------------------------------------------------- #
# TimeSeriesEstimator.predict, channels_last branch for DLWPFunctional models with insolation re-injection,
# DLWP/model/extensions.py:252-308 (loop) and :384-436 (output assembly).  xarray-free restatement: coordinates are plain arrays.
# ---------------------------------------------------------------------------------------------------------------- #

def estimator_rollout_ref(predict_fn, predictors, steps, insolation_rows, start_index, n_steps, time_dim, its, ots,
                          constants=None, keep_time_dim=False, interval=1):
    """
    Restates the reference loop for a channels_last, non-recurrent generator (`_keep_time_axis` False) whose model outputs
    every input variable (its == ots):
      * `predict_fn(list_of_inputs) -> list of n_steps outputs`, each (B, *space, ots*V)            extensions.py:273
      * per sequence step the known insolation of the new times is appended as the last channel of every input time step
        (extensions.py:277-296) and the constants are re-attached (:304-306)
      * result (B, sequence_steps, n_steps, ...) -> (B, effective_steps, ...) (:308-310), then the f_hour assembly of :384-436.
    `insolation_rows(rows) -> (len(rows), *space)` plays `insolation(new_t + k dt, lat, lon)`; `start_index[b]` is the time row of
    sample b's first input step, so `new_t` advances by ots * n_steps rows per sequence step (:277).
    Returns (values, f_hour): values (f_hour, B, *space, V) or, with keep_time_dim, (f_hour, B, ots, *space, V).
    """
    if int(steps) < 1:
        raise ValueError('must use positive integer for steps')
    if ots > its:
        raise NotImplementedError
    es = ots
    effective_steps = int(np.ceil(steps / es))
    p = [np.array(a, dtype=np.float32) for a in predictors]
    B = p[0].shape[0]
    space = p[0].shape[1:-1]
    rank = len(space)
    sequence_steps = int(np.ceil(steps / n_steps / time_dim))
    out0 = predict_fn(p)
    t_shape = np.asarray(out0[0]).shape
    result = np.full((B, sequence_steps, n_steps) + tuple(t_shape[1:]), np.nan, dtype=np.float32)
    new_t = np.asarray(start_index, dtype=np.int64).copy()
    fwd_tr = (0, rank + 1) + tuple(range(1, 1 + rank)) + (-1,)
    bwd_tr = (0,) + tuple(range(2, 2 + rank)) + (1, -1)
    for s in range(sequence_steps):
        outs = out0 if s == 0 else predict_fn(p)
        result[:, s] = np.stack([np.asarray(o, dtype=np.float32) for o in outs], axis=1)
        new_t = new_t + ots * n_steps
        new_ins = [np.concatenate([np.expand_dims(insolation_rows(new_t + n + m * its)[:, None], axis=-1)
                                   for n in range(its)], axis=1) for m in range(n_steps)]
        r = result[:, s, -1].reshape(tuple(t_shape[:-1]) + (ots, -1)).transpose(fwd_tr)
        p = [np.concatenate([r, new_ins[0]], axis=-1).transpose(bwd_tr).reshape((B,) + tuple(space) + (-1,))] + new_ins[1:]
        if constants is not None:
            p.append(np.repeat(np.expand_dims(constants, axis=0), B, axis=0))
    n_dim_1 = result.size // int(np.prod(t_shape))
    result = result.reshape((t_shape[0], n_dim_1) + tuple(t_shape[1:]))[:, :effective_steps]
    rv = result.reshape((B, effective_steps) + tuple(space) + (ots, -1))
    if keep_time_dim:
        vals = rv.transpose((1, 0, -2) + tuple(range(2, 2 + rank)) + (-1,))
        f_hour = np.arange(1, effective_steps * (es + interval - 1) + 1, es + interval - 1)
        return vals, f_hour
    vals = rv.transpose((1, -2, 0) + tuple(range(2, 2 + rank)) + (-1,)).reshape(
        (rv.shape[1] * rv.shape[-2], rv.shape[0]) + tuple(space) + (-1,))
    f_hour = np.array([np.arange(0, es) + interval + e * (es - 1 + interval) for e in range(effective_steps)]).flatten()
    return vals[:steps], f_hour[:steps]
