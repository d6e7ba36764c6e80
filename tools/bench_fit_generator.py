#!/usr/bin/env python3
"""Generator-fed training rate: DLWPFunctional.fit_generator on an HBM-resident ArrayDataGenerator -- the way the reference's CS scripts
train (Azure/train_cs.py:433-446) -- for (a) unet2, 7 variables x 2 time steps (BASELINE config 3's network) and (b) the production wiring
(4 variables, insolation, constants, sequence = 2), C48, batch 32, bf16.  Prints samples/s next to bench.py's resident figures."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dlwp-cs_amd'))
import numpy as np, torch
from DLWP.keras import backend
from DLWP.model import DLWPFunctional
from DLWP.model.cs_unet import build_cs_model
from DLWP.model.generators import ArrayDataGenerator
backend.set_device('cuda:0')
N, T, B = 48, int(os.environ.get('T', 420)), 32
rng = np.random.default_rng(0)
for name, V, seq, with_forcing in (('unet2 (7 variables x 2 steps)', 7, None, False), ('production model (4 variables, solar, constants, sequence 2)', 4, 2, True)):
    arr = rng.standard_normal((T, V, 6, N, N)).astype(np.float32)
    sol = rng.random((T, 6, N, N)).astype(np.float32) if with_forcing else None
    const = rng.standard_normal((2, 6, N, N)).astype(np.float32) if with_forcing else None
    dlwp = DLWPFunctional(is_convolutional=True, time_dim=2)
    gen = ArrayDataGenerator(dlwp, arr, rank=3, batch_size=B, input_time_steps=2, output_time_steps=2, sequence=seq,
                             insolation_array=sol, constants=const, channels_last=True, shuffle=True, device=True, dtype='bfloat16')
    backend.set_compute_dtype('bfloat16')
    try:
        model = build_cs_model(gen.convolution_shape, 2 * V, 'unet2', base_filter_number=32, integration_steps=seq or 1, io_time_steps=2,
                               insolation_shape=gen.insolation_shape if with_forcing else None,
                               constants_shape=(6, N, N, 2) if with_forcing else None)
    finally:
        backend.set_compute_dtype('float32')
    kw = dict(loss_weights=[0.5, 0.5]) if seq else {}
    dlwp.build_model(model, loss='mse', optimizer='adam', **kw)
    dlwp.fit_generator(gen, epochs=1, verbose=0)
    torch.cuda.synchronize()
    dlwp.fit_generator(gen, epochs=2, verbose=0)
    torch.cuda.synchronize()
    t0 = time.time()
    ep = 3
    dlwp.fit_generator(gen, epochs=ep, verbose=0)
    torch.cuda.synchronize()
    dt = time.time() - t0
    steps = ep * len(gen)
    print('%s: %d steps of batch %d in %.2f s = %.0f samples/s (%.3f ms per step)' % (name, steps, B, dt, steps * B / dt, 1e3 * dt / steps))
