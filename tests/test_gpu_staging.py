"""
GPU tests (`-m gpu`) of the host -> HBM feed of Model.fit (DLWP/keras/staging.py): batches staged ahead of the training step
through pinned memory and a copy stream must train EXACTLY like batches uploaded one by one -- shuffled and unshuffled epochs, float64 sources,
epochs cut short by steps_per_epoch, several inputs of one shape, and sources that are not arrays (generator objects).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), 'GPU tests need a HIP device'
    return torch.device('cuda', 0)


def _model(C=3):
    from DLWP.keras import backend
    from DLWP.model.cs_unet import build_cs_model
    backend.set_device('cuda:0')
    np.random.seed(7)
    m = build_cs_model((6, 8, 8, C), C, 'unet2', base_filter_number=4)
    m.compile(optimizer='adam', loss='mse', metrics=['mae'])
    return m


def _flat(model):
    return np.concatenate([w.ravel() for w in model.get_weights()])


def _fit(staged, x, y, C=3, **kw):
    _dev()
    os.environ['DLWPCS_OPTIONS'] = 'host_staging=%d' % (1 if staged else 0)
    try:
        m = _model(C)
        np.random.seed(11)                       # the epoch shuffles
        h = m.fit(x, y, verbose=0, **kw)
        torch.cuda.synchronize()
        return _flat(m), h.history
    finally:
        os.environ.pop('DLWPCS_OPTIONS', None)


@pytest.mark.parametrize('shuffle', [False, True])
@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_staged_fit_equals_plain_fit(shuffle, dtype):
    rng = np.random.default_rng(3)
    x = rng.standard_normal((22, 6, 8, 8, 3)).astype(dtype)      # 22 = 5 batches of 4 + a ragged one of 2
    y = rng.standard_normal((22, 6, 8, 8, 3)).astype(dtype)
    a, ha = _fit(True, x, y, batch_size=4, epochs=3, shuffle=shuffle)
    b, hb = _fit(False, x, y, batch_size=4, epochs=3, shuffle=shuffle)
    assert np.array_equal(a, b)
    assert ha == hb


def test_staged_fit_with_an_epoch_cut_short_and_a_strided_source():
    rng = np.random.default_rng(4)
    big = rng.standard_normal((40, 6, 8, 8, 6)).astype(np.float32)
    x, y = big[..., :3], big[..., 3:]                              # non-contiguous views of one array
    a, _ = _fit(True, x, y, batch_size=4, epochs=4, shuffle=True, steps_per_epoch=3)
    b, _ = _fit(False, x, y, batch_size=4, epochs=4, shuffle=True, steps_per_epoch=3)
    assert np.array_equal(a, b)


def test_staged_fit_from_a_sequence_of_batches():
    """keras.utils.Sequence-like source (what DLWP's generators are): items are (inputs, targets) of host arrays"""
    rng = np.random.default_rng(5)
    xs = [rng.standard_normal((4, 6, 8, 8, 3)).astype(np.float32) for _ in range(6)]
    ys = [rng.standard_normal((4, 6, 8, 8, 3)).astype(np.float32) for _ in range(6)]

    class Seq(object):
        def __len__(self):
            return len(xs)

        def __getitem__(self, i):
            return xs[i], ys[i]
    a, _ = _fit(True, Seq(), None, epochs=2)
    b, _ = _fit(False, Seq(), None, epochs=2)
    assert np.array_equal(a, b)


def test_stager_reuses_its_pinned_buffers():
    from DLWP.keras.staging import LazyTake, Stager
    st = Stager(_dev())
    rng = np.random.default_rng(6)
    src = rng.standard_normal((64, 5, 7)).astype(np.float64)
    for k in range(40):
        sel = rng.permutation(64)[:8]
        (d,), ev = st.upload([LazyTake(src, sel)], [torch.float32])
        torch.cuda.current_stream().wait_event(ev)
        assert np.array_equal(d.cpu().numpy(), src[sel].astype(np.float32))
    assert len(st._rings[(8, 5, 7)]) <= 4


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
def test_staged_predict_equals_plain_predict(dtype):
    from DLWP.keras import backend
    from DLWP.model.cs_unet import build_cs_model
    _dev()
    backend.set_device('cuda:0')
    backend.set_compute_dtype(dtype)
    try:
        np.random.seed(9)
        m = build_cs_model((6, 8, 8, 3), 3, 'unet2', base_filter_number=4)
    finally:
        backend.set_compute_dtype('float32')
    rng = np.random.default_rng(8)
    x = rng.standard_normal((23, 6, 8, 8, 3))           # float64, ragged last batch
    res = []
    for staged in ('1', '0'):
        os.environ['DLWPCS_OPTIONS'] = 'host_staging=' + staged
        try:
            res.append(m.predict(x, batch_size=4))
        finally:
            os.environ.pop('DLWPCS_OPTIONS', None)
    assert res[0].shape == (23, 6, 8, 8, 3) and res[0].dtype == np.float32
    assert np.array_equal(res[0], res[1])


def test_staged_evaluate_equals_plain_evaluate():
    rng = np.random.default_rng(10)
    x = rng.standard_normal((14, 6, 8, 8, 3)).astype(np.float32)
    y = rng.standard_normal((14, 6, 8, 8, 3)).astype(np.float32)
    _dev()
    m = _model(3)
    res = []
    for staged in ('1', '0'):
        os.environ['DLWPCS_OPTIONS'] = 'host_staging=' + staged
        try:
            res.append(m.evaluate(x, y, batch_size=4, verbose=0))
        finally:
            os.environ.pop('DLWPCS_OPTIONS', None)
    assert res[0] == res[1]


def test_device_generator_in_another_dtype_trains_like_the_host_path():
    """device-resident fp32 batches fed to a bf16 model (the reference's order: generator first, mixed precision
    switched on at compile time) must be cast on the COMPUTE stream behind their producer, not on the copy stream: the
    result has to equal the host-fed run bit for bit, also when the batch tensors are produced right before every step."""
    from DLWP.keras import backend
    from DLWP.model.cs_unet import build_cs_model
    dev = _dev()
    backend.set_device('cuda:0')
    rng = np.random.default_rng(12)
    x = rng.standard_normal((12, 6, 8, 8, 4)).astype(np.float32)
    y = rng.standard_normal((12, 6, 8, 8, 4)).astype(np.float32)

    class DeviceBatches(object):
        """Sequence-like: every item is assembled on the device by kernels of the compute stream (fp32)"""
        def __len__(self):
            return 3

        def __getitem__(self, i):
            xb = torch.from_numpy(x[4 * i:4 * i + 4]).to(dev)
            yb = torch.from_numpy(y[4 * i:4 * i + 4]).to(dev)
            for _ in range(20):                       # keep the producer busy on the compute stream
                xb = xb * 1.0
            return [xb], [yb]

    res = []
    for feed in ('host', 'device'):
        backend.set_compute_dtype('bfloat16')
        try:
            np.random.seed(7)
            m = build_cs_model((6, 8, 8, 4), 4, 'unet2', base_filter_number=4)
        finally:
            backend.set_compute_dtype('float32')
        m.compile(optimizer='adam', loss='mse', metrics=['mae'])
        if feed == 'host':
            m.fit(x, y, batch_size=4, epochs=2, shuffle=False, verbose=0)
        else:
            m.fit(DeviceBatches(), epochs=2, verbose=0)
        torch.cuda.synchronize()
        res.append(_flat(m))
        p = m.predict(x[:4] if feed == 'host' else torch.from_numpy(x[:4]).to(dev))
        res.append(np.asarray(p))
    assert np.array_equal(res[0], res[2])
    assert np.array_equal(res[1], res[3])


def test_multi_epoch_fit_keeps_the_data_in_hbm_and_trains_the_same():
    """fit() on host arrays over several epochs (engine option resident_data): the first epoch's uploads are kept in device buffers,
    later epochs gather their batches out of HBM.  Same shuffles (one numpy draw per epoch either way), same batches, same kernels:
    the weights after three shuffled epochs are bitwise those of the all-host-fed run; the second epoch on does not touch the link."""
    import os
    from DLWP.keras import backend
    from DLWP.model.cs_unet import build_cs_model
    backend.set_device('cuda:0')
    N, C, n = 16, 6, 40                          # 40 samples, batches of 16: a ragged last batch
    rng = np.random.default_rng(4)
    x = rng.standard_normal((n, 6, N, N, C)).astype(np.float32)
    y = rng.standard_normal((n, 6, N, N, C)).astype(np.float32)
    out = []
    for on in ('0', '1'):
        os.environ['DLWPCS_OPTIONS'] = 'resident_data=' + on
        try:
            backend.set_compute_dtype('bfloat16')
            try:
                np.random.seed(5)
                model = build_cs_model((6, N, N, C), C, 'unet2', base_filter_number=8)
            finally:
                backend.set_compute_dtype('float32')
            model.compile(optimizer='adam', loss='mse', metrics=['mae'])
            np.random.seed(11)
            hist = model.fit(x, y, batch_size=16, epochs=3, verbose=0, shuffle=True)
            torch.cuda.synchronize()
            res = model._last_resident
            assert (res is not None and res['state'] == 'ready' and res['bufs'] is None) == (on == '1')
            out.append((np.concatenate([w.ravel() for w in model.get_weights()]), np.array(hist.history['loss'])))
        finally:
            os.environ.pop('DLWPCS_OPTIONS', None)
    assert np.array_equal(out[0][0], out[1][0])
    assert np.array_equal(out[0][1], out[1][1])
    # one epoch, or a bounded number of steps per epoch: nothing to keep
    model.fit(x, y, batch_size=16, epochs=1, verbose=0)
    assert model._last_resident is None
    model.fit(x, y, batch_size=16, epochs=3, steps_per_epoch=2, verbose=0)
    assert model._last_resident is None
