#!/usr/bin/env python3
"""Host-fed training rate: Model.fit on numpy arrays that live in host memory (the reference's own feeding mode), unet2 at
C48 / 14 channels / batch 32 in bf16.  Prints steps/s and the device-resident rate of bench.py's configuration next to it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dlwp-cs_amd'))
import numpy as np, torch
from DLWP.keras import backend
from DLWP.model.cs_unet import build_cs_model
backend.set_device('cuda:0')
backend.set_compute_dtype('bfloat16')
model = build_cs_model((6, 48, 48, 14), 14, 'unet2', base_filter_number=32)
backend.set_compute_dtype('float32')
model.compile(optimizer='adam', loss='mse')
n = 32 * int(os.environ.get('NB', 16))
rng = np.random.default_rng(0)
x = rng.standard_normal((n, 6, 48, 48, 14)).astype(np.float32)
y = rng.standard_normal((n, 6, 48, 48, 14)).astype(np.float32)
model.fit(x, y, batch_size=32, epochs=2, verbose=0, shuffle=False)
torch.cuda.synchronize()
for shuffle in (False, True):
    t0 = time.time()
    ep = 4
    model.fit(x, y, batch_size=32, epochs=ep, verbose=0, shuffle=shuffle)
    torch.cuda.synchronize()
    dt = time.time() - t0
    steps = ep * n // 32
    print('host-fed fit, shuffle=%s: %.1f steps/s = %.0f samples/s (%.2f ms per step; batch = %.1f MB of fp32 x + y)'
          % (shuffle, steps / dt, steps * 32 / dt, 1e3 * dt / steps, 2 * x[:32].nbytes / 1e6))
for stg in ('1', '0'):
    os.environ['DLWPCS_OPTIONS'] = 'host_staging=' + stg
    model.predict(x[:64], batch_size=32)
    torch.cuda.synchronize()
    t0 = time.time()
    out = model.predict(x, batch_size=32)
    dt = time.time() - t0
    print('host-fed predict, staging=%s: %.0f samples/s (%.2f ms per batch of 32; 24.8 MB up, 24.8 MB down)' % (stg, n / dt, 1e3 * dt / (n // 32)))
