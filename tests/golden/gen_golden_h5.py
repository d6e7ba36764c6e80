#!/usr/bin/env python3
"""
HDF5 fixtures for the keras-file import (reference DLWP/util.py:139-142,157-193, DLWP/custom.py:184-191).

TensorFlow is not installable here, but the build container carries h5py 3.3 / libhdf5 1.10.6 under /opt/conda (python3.9).
This script writes, WITH THE REAL HDF5 LIBRARY, files in exactly the layout of keras 2.x (`save_weights_to_hdf5_group` /
`save_model_to_hdf5`: root or `model_weights` group with attribute `layer_names`, one group per layer -- also the weightless
ones -- with attribute `weight_names`, one float32 dataset per weight named `<layer>/<weight>:0`, attributes `backend`,
`keras_version`, `model_config`, `training_config` as byte strings):

  tests/golden/h5_weights_tiny.h5   model.save_weights(..., save_format='h5') of a tiny unet2 (N = 8, 3 channels, base 4)
  tests/golden/h5_model_tiny.h5     model.save(...) of the same model (config JSON + weights + Adam training config)
  tests/golden/h5_expected.npz      the weight arrays in keras order, for the reader test

The engine's pure-Python reader (DLWP/keras/hdf5_lite.py) is tested against these on every box (no h5py needed to READ).
Run:  python tests/golden/gen_golden_h5.py        (system python; shells out to /opt/conda/bin/python3.9 for the h5py part)
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'dlwp-cs_amd'))
H5PY_PYTHON = '/opt/conda/bin/python3.9'

WRITER = r'''
import json, sys
import numpy as np
import h5py
spec = json.load(open(sys.argv[1]))
arrs = np.load(sys.argv[2])

FIXED = True

def save_attributes(group, name, data):
    # keras.engine.saving.save_attributes_to_hdf5_group (attributes below the 64 KB object-header limit).  h5py 2.10 (the
    # TensorFlow 2.1 environment of the reference, environment.yml) stored lists of bytes as FIXED-length strings (numpy 'S'
    # arrays); h5py >= 3 stores them as variable-length strings.  The weights file is written the old way, the model file
    # the new way, so the reader is exercised on both.
    group.attrs[name] = np.array(data) if (FIXED and len(data)) else data

def scalar(b):
    return np.bytes_(b) if FIXED else b

def save_weights(group, layers):
    save_attributes(group, 'layer_names', [l['name'].encode('utf8') for l in layers])
    group.attrs['backend'] = scalar('tensorflow'.encode('utf8'))
    group.attrs['keras_version'] = scalar('2.2.4-tf'.encode('utf8'))
    for l in layers:
        g = group.create_group(l['name'])
        names = [n.encode('utf8') for n in l['weights']]
        save_attributes(g, 'weight_names', names)
        for n in l['weights']:
            val = arrs[spec['keys'][n]]
            d = g.create_dataset(n, val.shape, dtype=val.dtype)
            if not val.shape:
                d[()] = val
            else:
                d[:] = val

with h5py.File(spec['weights_file'], 'w') as f:
    save_weights(f, spec['layers'])
FIXED = False
with h5py.File(spec['model_file'], 'w') as f:
    f.attrs['keras_version'] = '2.2.4-tf'.encode('utf8')
    f.attrs['backend'] = 'tensorflow'.encode('utf8')
    f.attrs['model_config'] = np.bytes_(json.dumps(spec['model_config']).encode('utf8'))      # fixed-length, ~10 KB
    save_weights(f.create_group('model_weights'), spec['layers'])
    f.attrs['training_config'] = json.dumps(spec['training_config']).encode('utf8')
    f.attrs['note_vlen'] = 'a variable-length string attribute (newer keras / h5py write these)'
    f.attrs['numbers'] = np.arange(5, dtype=np.int32)
print('h5py', h5py.__version__, 'hdf5', h5py.version.hdf5_version)
'''


def main():
    from DLWP.keras import backend
    backend.set_device('cpu')
    from DLWP.keras.engine import reset_uids
    from DLWP.model.cs_unet import build_cs_model
    reset_uids()
    np.random.seed(11)
    model = build_cs_model((6, 8, 8, 3), 3, 'unet2', base_filter_number=4)
    # biases are zero-initialised: make every array distinctive
    rng = np.random.default_rng(12)
    model.set_weights([w + rng.standard_normal(w.shape).astype(np.float32) * 0.1 for w in model.get_weights()])
    layers, keys, arrays = [], {}, {}
    for lay in model.layers:
        names = list(lay._weight_names)
        for n, a in zip(names, lay.get_weights()):
            keys[n] = 'w%03d' % len(keys)
            arrays[keys[n]] = a
        layers.append({'name': lay.name, 'weights': names})
    training = {'optimizer_config': {'class_name': 'Adam', 'config': {'name': 'Adam', 'learning_rate': 0.002, 'decay': 0.0,
                                                                    'beta_1': 0.9, 'beta_2': 0.999, 'epsilon': 1e-07,
                                                                    'amsgrad': False}},
                'loss': 'mse', 'metrics': ['mae'], 'weighted_metrics': None, 'sample_weight_mode': None,
                'loss_weights': None}
    spec = {'layers': layers, 'keys': keys, 'model_config': model.to_keras_config(), 'training_config': training,
            'weights_file': os.path.join(HERE, 'h5_weights_tiny.h5'), 'model_file': os.path.join(HERE, 'h5_model_tiny.h5')}
    with tempfile.TemporaryDirectory() as tmp:
        sp, ap, wp = os.path.join(tmp, 'spec.json'), os.path.join(tmp, 'arrays.npz'), os.path.join(tmp, 'writer.py')
        json.dump(spec, open(sp, 'w'))
        np.savez(ap, **arrays)
        open(wp, 'w').write(WRITER)
        out = subprocess.run([H5PY_PYTHON, wp, sp, ap], capture_output=True, text=True)
        print(out.stdout, out.stderr[-2000:])
        out.check_returncode()
    np.savez_compressed(os.path.join(HERE, 'h5_expected.npz'), names=np.array([n for l in layers for n in l['weights']]),
                        layer_names=np.array([l['name'] for l in layers]),
                        **{'w%03d' % i: arrays[keys[n]] for i, n in enumerate(n for l in layers for n in l['weights'])})
    print('layers:', [l['name'] for l in layers])


if __name__ == '__main__':
    main()
