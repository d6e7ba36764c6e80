"""
N3: files written by keras + h5py load into the engine without h5py / TensorFlow (DLWP/keras/hdf5_lite.py).
Fixtures tests/golden/h5_*.h5 were written by the real HDF5 library (h5py 3.3 / libhdf5 1.10.6) in the layout of keras'
save_weights / save (tests/golden/gen_golden_h5.py): the weights file with fixed-length string attributes (h5py 2.10, the
reference's TensorFlow 2.1 environment), the model file with variable-length ones (h5py >= 3).
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import cs_oracle as orc


@pytest.fixture(autouse=True)
def _cpu_device():
    from DLWP.keras import backend
    backend.set_device('cpu')
    yield


def _expected(golden_dir):
    e = np.load(os.path.join(golden_dir, 'h5_expected.npz'))
    names = [str(n) for n in e['names']]
    return names, [e['w%03d' % i] for i in range(len(names))], [str(n) for n in e['layer_names']]


def _tiny(reset=True):
    from DLWP.keras.engine import reset_uids
    from DLWP.model.cs_unet import build_cs_model
    if reset:
        reset_uids()
    np.random.seed(99)
    return build_cs_model((6, 8, 8, 3), 3, 'unet2', base_filter_number=4)


@pytest.mark.parametrize('fname', ['h5_weights_tiny.h5', 'h5_model_tiny.h5'])
def test_hdf5_reader_against_libhdf5_files(golden_dir, fname):
    from DLWP.keras import hdf5_lite as h5
    path = os.path.join(golden_dir, fname)
    assert h5.is_hdf5(path) and not h5.is_hdf5(os.path.join(golden_dir, 'h5_expected.npz'))
    names, arrays, layer_names = _expected(golden_dir)
    layers, cfg = h5.read_keras_weights(path)
    assert [n for n, _ in layers] == layer_names                      # every layer has a group, weightless ones too
    flat = [(n, a) for _, ws in layers for n, a in ws]
    assert [n for n, _ in flat] == names
    for (n, a), ref in zip(flat, arrays):
        assert a.dtype == np.float32 and a.shape == ref.shape and np.array_equal(a, ref), n
    f = h5.File(path)
    root = f['model_weights'] if fname == 'h5_model_tiny.h5' else f
    assert h5._as_str(root.attrs['backend']) == 'tensorflow' and h5._as_str(root.attrs['keras_version']) == '2.2.4-tf'
    g = root['cube_sphere_conv2d']
    assert sorted(g.keys()) == ['cube_sphere_conv2d'] and len(g['cube_sphere_conv2d'].keys()) == 4
    d = root['output/output/polar_bias:0']
    assert d.shape == (3,) and np.array_equal(d[:2], arrays[names.index('output/polar_bias:0')][:2])
    assert root['re_lu'].attrs['weight_names'].shape == (0,)
    with pytest.raises(KeyError):
        root['no_such_layer']
    if fname == 'h5_model_tiny.h5':
        assert cfg is not None and json.loads(cfg)['class_name'] == 'Model'
        assert h5._as_str(f.attrs['note_vlen']).startswith('a variable-length string')
        assert np.array_equal(f.attrs['numbers'], np.arange(5, dtype=np.int32))
    else:
        assert cfg is None


def test_load_weights_from_keras_hdf5(golden_dir):
    names, arrays, _ = _expected(golden_dir)
    model = _tiny()
    model.load_weights(os.path.join(golden_dir, 'h5_weights_tiny.h5'))
    for a, ref in zip(model.get_weights(), arrays):
        assert np.array_equal(a, ref)
    # by_name: layer names of a model built in the same order match keras' auto-naming
    m2 = _tiny()
    m2.load_weights(os.path.join(golden_dir, 'h5_model_tiny.h5'), by_name=True)
    assert [n for l in m2._weight_layers() for n in l._weight_names] == names
    for a, ref in zip(m2.get_weights(), arrays):
        assert np.array_equal(a, ref)
    # a model with different layer names still loads in topological order, and by_name then matches nothing
    m3 = _tiny(reset=False)
    before = m3.get_weights()
    m3.load_weights(os.path.join(golden_dir, 'h5_weights_tiny.h5'), by_name=True)
    after = m3.get_weights()
    assert all(np.array_equal(a, b) for a, b in zip(after[:-4], before[:-4]))           # auto-named layers: no match
    assert all(np.array_equal(a, b) for a, b in zip(after[-4:], arrays[-4:]))           # the layer named 'output' matched
    m3.load_weights(os.path.join(golden_dir, 'h5_weights_tiny.h5'))
    assert all(np.array_equal(a, b) for a, b in zip(m3.get_weights(), arrays))
    # wrong architecture -> keras' error
    from DLWP.model.cs_unet import build_cs_model
    with pytest.raises(ValueError, match='weight file containing'):
        build_cs_model((6, 8, 8, 3), 3, 'basic', base_filter_number=4).load_weights(
            os.path.join(golden_dir, 'h5_weights_tiny.h5'))


def test_load_model_from_keras_hdf5(golden_dir):
    from DLWP.keras.models import load_model
    names, arrays, layer_names = _expected(golden_dir)
    model = load_model(os.path.join(golden_dir, 'h5_model_tiny.h5'))
    assert [l.name for l in model.layers] == layer_names
    assert model.n_fused == _tiny().n_fused == 10 and model.count_params() == sum(a.size for a in arrays)
    for a, ref in zip(model.get_weights(), arrays):
        assert np.array_equal(a, ref)
    assert model._compiled and model.optimizer.learning_rate == 0.002 and model.metrics == ['mae'] and model.loss == 'mse'
    ref = _tiny()
    assert json.dumps(model.to_keras_config(), sort_keys=True) == json.dumps(ref.to_keras_config(), sort_keys=True)
    with pytest.raises(ValueError, match='holds weights only'):
        load_model(os.path.join(golden_dir, 'h5_weights_tiny.h5'))


def test_native_container_roundtrip_has_no_pickle(tmp_path):
    from DLWP.keras import serialization
    from DLWP.keras.models import load_model
    model = _tiny()
    model.compile(optimizer='adam', loss='mse', metrics=['mae'])
    w = str(tmp_path / 'weights.ckpt')                  # no HDF5 extension, no save_format: the native container
    model.save_weights(w)
    with open(w, 'rb') as f:
        assert f.read(2) == b'PK'                       # zip container, not a pickle, not HDF5
    arrays, meta = serialization.load_container(w)
    assert meta['format'] == 'dlwpcs-weights-2' and len(arrays) == len(model.weights)
    m = str(tmp_path / 'm.model')
    model.save(m, save_format='npz')
    with open(m, 'rb') as f:
        assert f.read(2) == b'PK'
    loaded = load_model(m)
    assert all(np.array_equal(a, b) for a, b in zip(model.get_weights(), loaded.get_weights()))
    assert loaded._compiled and loaded.metrics == ['mae']
    bad = str(tmp_path / 'old.pkl')
    import pickle
    with open(bad, 'wb') as f:
        pickle.dump({'format': 'dlwpcs-weights-1'}, f)
    with pytest.raises(ValueError, match='round-1 pickle'):
        model.load_weights(bad)


@pytest.mark.gpu
def test_model_loaded_from_keras_hdf5_predicts_like_the_oracle(golden_dir):
    from DLWP.keras import backend
    backend.set_device('cuda:0')
    from DLWP.keras.models import load_model
    names, arrays, _ = _expected(golden_dir)
    model = load_model(os.path.join(golden_dir, 'h5_model_tiny.h5'))
    params = []
    for i in range(0, len(arrays), 4):
        params.append({'equatorial_kernel': torch.tensor(arrays[i], dtype=torch.float64),
                       'polar_kernel': torch.tensor(arrays[i + 1], dtype=torch.float64),
                       'equatorial_bias': torch.tensor(arrays[i + 2], dtype=torch.float64),
                       'polar_bias': torch.tensor(arrays[i + 3], dtype=torch.float64)})
    x = np.random.default_rng(4).standard_normal((2, 6, 8, 8, 3)).astype(np.float32)
    y = model.predict(x)
    ref = orc.unet2_forward(torch.tensor(x, dtype=torch.float64), params).numpy()
    assert np.abs(y - ref).max() <= 1e-5 * np.abs(ref).max()


def test_util_load_model_reads_the_reference_file_triple(golden_dir, tmp_path):
    """<name>.keras written by keras under TensorFlow (HDF5 layout), <name>.pkl / <name>.history pickled by the REFERENCE's
    own DLWPFunctional / save_model (tests/golden/gen_golden_rollout.py: gen_wrapper): DLWP.util.load_model of the engine
    returns a working wrapper (reference DLWP/util.py:157-193)."""
    import shutil
    from DLWP.model import DLWPFunctional
    from DLWP.util import is_channels_last, load_model
    base = str(tmp_path / 'dlwp_model')
    shutil.copy(os.path.join(golden_dir, 'h5_model_tiny.h5'), base + '.keras')
    shutil.copy(os.path.join(golden_dir, 'ref_wrapper.pkl'), base + '.pkl')
    shutil.copy(os.path.join(golden_dir, 'ref_wrapper.history'), base + '.history')
    dlwp, hist = load_model(base, history=True)
    assert isinstance(dlwp, DLWPFunctional)
    assert (dlwp.time_dim, dlwp._n_steps, dlwp.is_convolutional, dlwp.is_recurrent, dlwp.impute, dlwp.FHW_DIMS) == \
        (2, 2, True, False, False, True)
    assert dlwp.scaler is None and dlwp.gpus == 1
    assert hist == {'loss': [1.5, 0.75], 'val_loss': [1.25, 0.875]}
    names, arrays, _ = _expected(golden_dir)
    assert dlwp.model is dlwp.base_model and dlwp.model._compiled
    assert all(np.array_equal(a, b) for a, b in zip(dlwp.model.get_weights(), arrays))
    assert is_channels_last(dlwp)
