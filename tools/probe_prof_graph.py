"""Where does the in-graph profiler fail?  (GPU box)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'dlwp-cs_amd'))
import numpy as np, torch
from DLWP.keras import backend
from DLWP.model.cs_unet import build_cs_model
from DLWP import _native as nat
backend.set_device('cuda:0')
dev = torch.device('cuda', 0)
lib = nat.lib()
rng = np.random.default_rng(2)
x = torch.tensor(rng.standard_normal((4, 6, 16, 16, 6)), dtype=torch.float32, device=dev)
t = torch.tensor(rng.standard_normal((4, 6, 16, 16, 6)), dtype=torch.float32, device=dev)
m = build_cs_model((6, 16, 16, 6), 6, 'unet2', base_filter_number=8)
m.compile(optimizer='adam', loss='mse')
lib.dlwpcs_prof_reset(); lib.dlwpcs_prof_enable(1)
print('current stream', torch.cuda.current_stream().cuda_stream)
for i in range(4):
    try:
        m.train_on_device_batch([x], [t]); torch.cuda.synchronize()
        print('step', i, 'ok, graphs', len(m._graphs), 'records', lib.dlwpcs_prof_count())
    except Exception as e:
        print('step', i, 'FAILED', type(e).__name__, e)
        break
import ctypes
tag = ctypes.create_string_buffer(160)
ms, fl, by = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
lib.dlwpcs_prof_enable(0)
n = lib.dlwpcs_prof_count()
for r in range(2):
    m.train_on_device_batch([x], [t]); torch.cuda.synchronize()
    tot, ok, bad = 0.0, 0, 0
    for i in range(n):
        rc = lib.dlwpcs_prof_get(i, tag, 160, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by))
        name = tag.value.decode()
        if rc != 0:
            bad += 1
            continue
        if name.endswith('@graph'):
            ok += 1; tot += ms.value
            if r == 1: print('  %-100s %.2f us' % (name[:100], 1e3 * ms.value))
    print('replay', r, 'graph records ok', ok, 'failed queries', bad, 'sum %.1f us' % (1e3 * tot))
