"""
The engine WRITES Keras-layout HDF5 (DLWP.keras.hdf5_lite.write_keras_file) for what the reference saves as HDF5:
`model.save_weights(path, save_format='h5')` (DLWP/custom.py:186, the checkpoint callback) and `model.save('<name>.keras')`
(DLWP/util.py:139, save_model).  Checked (1) through the engine's own reader -- which is pinned to files written by the real HDF5
library (tests/test_h5_import.py) -- and (2), where the build container carries h5py + libhdf5 (/opt/conda, python3.9; not on the GPU
boxes), by opening the files with the real library: structure, attribute types, data, and an append by libhdf5 into the file.
"""
import json
import os
import subprocess

import numpy as np
import pytest

H5PY_PYTHON = '/opt/conda/bin/python3.9'


def _has_h5py():
    if not os.path.exists(H5PY_PYTHON):
        return False
    return subprocess.run([H5PY_PYTHON, '-c', 'import h5py'], capture_output=True).returncode == 0


def _tiny():
    from DLWP.model.cs_unet import build_cs_model
    np.random.seed(11)
    return build_cs_model((6, 8, 8, 3), 3, 'unet2', base_filter_number=4)


def test_weights_and_model_files_round_trip_through_the_reader(tmp_path):
    from DLWP.keras import hdf5_lite
    from DLWP.keras.models import load_model
    model = _tiny()
    model.compile(optimizer='adam', loss='mse', metrics=['mae'])
    w = str(tmp_path / 'weights.h5')
    model.save_weights(w, save_format='h5')             # what the reference's callback calls (custom.py:186)
    assert hdf5_lite.is_hdf5(w)
    layers, cfg = hdf5_lite.read_keras_weights(w)
    assert cfg is None and [n for n, _ in layers] == [l.name for l in model.layers]        # weightless layers have groups too
    flat = [a for _, ws in layers for _, a in ws]
    assert len(flat) == len(model.weights) and all(np.array_equal(a, b) for a, b in zip(flat, model.get_weights()))
    other = _tiny()
    other.set_weights([np.zeros_like(a) for a in other.get_weights()])
    other.load_weights(w)
    assert all(np.array_equal(a, b) for a, b in zip(other.get_weights(), model.get_weights()))
    # model file under the reference's name: graph + weights + compile arguments
    m = str(tmp_path / 'net.keras')
    model.save(m)
    assert hdf5_lite.is_hdf5(m)
    loaded = load_model(m)
    assert [l.name for l in loaded.layers] == [l.name for l in model.layers]
    assert all(np.array_equal(a, b) for a, b in zip(loaded.get_weights(), model.get_weights()))
    assert loaded._compiled and loaded.loss == 'mse' and loaded.metrics == ['mae']
    assert json.dumps(loaded.to_keras_config(), sort_keys=True) == json.dumps(model.to_keras_config(), sort_keys=True)


@pytest.mark.parametrize('n_links', [0, 1, 8, 9, 256, 257, 2500])
def test_groups_of_any_size_cost_their_own_links_only(tmp_path, n_links):
    """symbol nodes of <= 8 links under a multi-level v1 B-tree (libhdf5's default K): a 2500-layer file is ~4 MB, not 600"""
    from DLWP.keras import hdf5_lite as h5
    root = h5._WGroup()
    big = root.group('big')
    names = ['layer_%04d' % i for i in range(n_links)]
    rng = np.random.default_rng(n_links)
    for n in rng.permutation(names) if names else []:
        big.group(str(n)).children['w'] = np.full((3,), float(str(n)[-4:]), dtype=np.float32)
    root.group('small').children['x'] = np.arange(5, dtype=np.int64)
    path = str(tmp_path / 'big.h5')
    h5.write_tree(path, root)
    assert os.path.getsize(path) < 8192 + 1700 * n_links          # (each link here is a group of its own: ~1.3 KB)
    with h5.File(path) as f:
        assert sorted(f['big'].keys()) == names
        for n in names[::97]:
            assert np.array_equal(f['big'][n]['w'].read(), np.full((3,), float(n[-4:]), dtype=np.float32))
        assert np.array_equal(f['small']['x'].read(), np.arange(5))
    if _has_h5py():
        script = ('import h5py, sys\n'
                  'with h5py.File(sys.argv[1], "r") as f:\n'
                  '    ks = sorted(f["big"].keys())\n'
                  '    assert len(ks) == int(sys.argv[2]), len(ks)\n'
                  '    assert all(float(f["big"][k]["w"][0]) == float(k[-4:]) for k in ks)\n'
                  '    assert list(f["small"]["x"][:]) == [0, 1, 2, 3, 4]\n'
                  'with h5py.File(sys.argv[1], "a") as f:\n'
                  '    f["big"].create_group("zz_added")\n'
                  'with h5py.File(sys.argv[1], "r") as f:\n'
                  '    assert len(f["big"].keys()) == int(sys.argv[2]) + 1\n')
        r = subprocess.run([H5PY_PYTHON, '-c', script, path, str(n_links)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        with h5.File(path) as f:
            assert len(f['big'].keys()) == n_links + 1


def test_long_attributes_are_split_like_keras(tmp_path):
    from DLWP.keras import hdf5_lite
    names = ['layer_with_a_rather_long_name_%04d' % i for i in range(2500)]          # > 64 KB of layer names
    layers = [(n, []) for n in names[:-1]] + [(names[-1], [('%s/kernel:0' % names[-1], np.arange(6, dtype=np.float32).reshape(2, 3))])]
    p = str(tmp_path / 'big.h5')
    hdf5_lite.write_keras_file(p, layers)
    f = hdf5_lite.File(p)
    assert 'layer_names' not in f.attrs and 'layer_names0' in f.attrs and 'layer_names1' in f.attrs
    got, _ = hdf5_lite.read_keras_weights(p)
    assert [n for n, _ in got] == names and np.array_equal(got[-1][1][0][1], np.arange(6, dtype=np.float32).reshape(2, 3))


CHECK = r'''
import json, sys
import numpy as np
import h5py
path, expect = sys.argv[1], np.load(sys.argv[2])
out = {}
with h5py.File(path, 'r') as f:
    root = f['model_weights'] if 'model_weights' in f else f
    names = [n.decode('utf8') if isinstance(n, bytes) else n for n in root.attrs['layer_names']]
    out['layer_names'] = names
    out['backend'] = root.attrs['backend'].decode('utf8')
    k = 0
    ok = True
    for n in names:
        g = root[n]
        for w in g.attrs['weight_names']:                # keras' load_weights_from_hdf5_group reads exactly this
            w = w.decode('utf8') if isinstance(w, bytes) else w
            ok = ok and np.array_equal(np.asarray(g[w]), expect['a%d' % k])
            k += 1
    out['n_weights'], out['equal'] = k, bool(ok)
    if 'model_config' in f.attrs:
        mc = f.attrs['model_config']
        out['model_class'] = json.loads(mc.decode('utf8') if isinstance(mc, bytes) else mc)['class_name']
        tc = f.attrs['training_config']
        out['loss'] = json.loads(tc.decode('utf8') if isinstance(tc, bytes) else tc)['loss']
        ow = f['optimizer_weights']
        on = [w.decode('utf8') if isinstance(w, bytes) else w for w in ow.attrs['weight_names']]
        out['optimizer_weights'] = len(on)
        out['iter'] = int(np.asarray(ow[on[0]]))
# the real library can also WRITE into the file (heap / B-tree / free space are consistent)
with h5py.File(path, 'a') as f:
    f.create_dataset('added_by_libhdf5', data=np.arange(4.0))
    f.attrs['note'] = np.bytes_('ok')
with h5py.File(path, 'r') as f:
    out['after_append'] = sorted(f.keys())
print(json.dumps(out))
'''


@pytest.mark.skipif(not _has_h5py(), reason='no h5py / libhdf5 in this container')
def test_the_real_hdf5_library_reads_what_the_engine_writes(tmp_path):
    from DLWP.keras import hdf5_lite
    model = _tiny()
    model.compile(optimizer='adam', loss='mse', metrics=['mae'])
    model.optimizer._ensure_state(model._flat_params)
    arrays = model.get_weights()
    exp = str(tmp_path / 'expect.npz')
    np.savez(exp, **{'a%d' % i: a for i, a in enumerate(arrays)})
    script = str(tmp_path / 'check.py')
    open(script, 'w').write(CHECK)
    w, m = str(tmp_path / 'weights.h5'), str(tmp_path / 'net.keras')
    model.save_weights(w, save_format='h5')
    model.save(m)
    for path, is_model in ((w, False), (m, True)):
        r = subprocess.run([H5PY_PYTHON, script, path, exp], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        out = json.loads(r.stdout.strip().split('\n')[-1])
        assert out['layer_names'] == [l.name for l in model.layers] and out['backend'] == 'tensorflow'
        assert out['n_weights'] == len(arrays) and out['equal']
        assert 'added_by_libhdf5' in out['after_append']
        if is_model:
            assert out['model_class'] == 'Model' and out['loss'] == 'mse'
            assert out['optimizer_weights'] == 1 + 2 * len(arrays) and out['iter'] == 0
        # ... and the engine's reader still reads the file after libhdf5 has modified it
        layers, _ = hdf5_lite.read_keras_weights(path)
        assert all(np.array_equal(a, b) for a, b in zip([a for _, ws in layers for _, a in ws], arrays))


@pytest.mark.gpu
@pytest.mark.parametrize('fmt', ['keras', 'npz'])
def test_training_resumes_from_a_saved_model_file(tmp_path, fmt):
    """train 2 steps, save (`<name>.keras` = HDF5 with optimizer_weights, or the native container), load, train 1 more step:
    the same parameters as 3 uninterrupted steps -- the Adam moments and the step count travel in the file"""
    import torch
    from DLWP.keras import backend
    from DLWP.keras.models import load_model
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    backend.set_device('cuda:0')
    dev = torch.device('cuda', 0)
    rng = np.random.default_rng(5)
    x = torch.tensor(rng.standard_normal((2, 6, 8, 8, 3)), dtype=torch.float32, device=dev)
    t = torch.tensor(rng.standard_normal((2, 6, 8, 8, 3)), dtype=torch.float32, device=dev)
    a = _tiny()
    a.use_graphs = False
    a.compile(optimizer='adam', loss='mse')
    w0 = a.get_weights()
    for _ in range(2):
        a.train_on_device_batch([x], [t])
    path = str(tmp_path / ('net.keras' if fmt == 'keras' else 'net.model'))
    a.save(path, save_format=None if fmt == 'keras' else 'npz')
    a.train_on_device_batch([x], [t])
    b = load_model(path)
    b.use_graphs = False
    assert int(b.optimizer._step[0].item()) == 2
    b.train_on_device_batch([x], [t])
    torch.cuda.synchronize()
    for p, q in zip(a.get_weights(), b.get_weights()):
        assert np.array_equal(p, q)
    assert not np.array_equal(a.get_weights()[0], w0[0])
