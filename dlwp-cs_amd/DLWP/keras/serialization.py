"""
Native file container of the engine: an uncompressed numpy `.npz` archive (zip of .npy members) written under ANY file name
(`<name>.keras`, `weights.h5`, ... -- the reference scripts choose the names) holding an ordered list of arrays plus one JSON
metadata record.  Read back with `allow_pickle=False`: a model / weights file cannot carry executable content (the round-1
format was a pickle).
"""
import io
import json
import os

import numpy as np


def save_container(filepath, arrays, meta):
    buf = {'a%05d' % i: np.ascontiguousarray(a) for i, a in enumerate(arrays)}
    buf['meta_json'] = np.frombuffer(json.dumps(meta, default=lambda o: list(o)).encode('utf8'), dtype=np.uint8)
    tmp = '%s.tmp%d' % (filepath, os.getpid())
    with open(tmp, 'wb') as f:                 # a file object: numpy appends no '.npz' to the name
        np.savez(f, **buf)
    os.replace(tmp, filepath)


def load_container(filepath):
    with open(filepath, 'rb') as f:
        head = f.read(4)
    if head[:2] != b'PK':
        raise ValueError('%s is neither a dlwpcs container (npz) nor an HDF5 file%s' % (
            filepath, ' -- it looks like a round-1 pickle; re-save it with this version' if head[:1] == b'\x80' else ''))
    with np.load(filepath, allow_pickle=False) as z:
        keys = sorted(k for k in z.files if k.startswith('a') and k[1:].isdigit())
        arrays = [z[k] for k in keys]
        meta = json.loads(bytes(z['meta_json'].tobytes()).decode('utf8')) if 'meta_json' in z.files else {}
    return arrays, meta
