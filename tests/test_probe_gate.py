"""
The gather-form data gradient (csrc/conv_ws.h, EDGE) and the batched weight gradient mask operands by LDS ADDRESS: reads beyond
the workgroup's allocation must return zeros.  DLWP._native asks the device once (dlwpcs_lds_oob_probe) and decides; these tests
pin the DECISION on the host (the probe launch itself is replaced): pass -> gather form, fail -> padded grid + one warning, cached
per device, never inside a capture.  (Adjoint of /root/reference/DLWP/custom.py:1198-1308 either way: only the kernel form changes.)
"""
import warnings

import pytest

from DLWP import _native as nat


@pytest.fixture
def gate(monkeypatch):
    calls = []
    state = {'bad': 0, 'capturing': False}

    def probe(device):
        calls.append(str(device))
        return state['bad']

    monkeypatch.setattr(nat, '_run_lds_oob_probe', probe)
    monkeypatch.setattr(nat, '_capturing', lambda: state['capturing'])
    monkeypatch.setattr(nat, 'halo_tables', lambda N, p, device: None)
    monkeypatch.setattr(nat, '_lds_probe', {})
    monkeypatch.setattr(nat, '_gather_ok', {(48, 1, 'cuda:0'), (24, 1, 'cuda:0'), (48, 1, 'cuda:1')})
    return state, calls


def test_probe_pass_enables_gather_form_once_per_device(gate):
    state, calls = gate
    assert nat.dgrad_gather_ready(48, 1, 'cuda:0')
    assert nat.dgrad_gather_ready(24, 1, 'cuda:0')
    assert nat.dgrad_gather_ready(48, 1, 'cuda:1')
    assert calls == ['cuda:0', 'cuda:1']            # one launch per device, not per table
    assert not nat.dgrad_gather_ready(12, 1, 'cuda:0')          # no plan buffer for this size: padded grid, no probe needed
    assert calls == ['cuda:0', 'cuda:1']


def test_probe_failure_falls_back_with_one_warning(gate):
    state, calls = gate
    state['bad'] = 7
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        assert not nat.dgrad_gather_ready(48, 1, 'cuda:0')
        assert not nat.dgrad_gather_ready(24, 1, 'cuda:0')
        assert not nat.lds_oob_reads_zero('cuda:0')
    assert len([x for x in w if 'dlwpcs_lds_oob_probe' in str(x.message)]) == 1
    assert calls == ['cuda:0']
    # another device is asked on its own
    state['bad'] = 0
    assert nat.dgrad_gather_ready(48, 1, 'cuda:1')


def test_first_use_inside_a_capture_refuses(gate):
    state, calls = gate
    state['capturing'] = True
    with pytest.raises(nat.NativeError):
        nat.lds_oob_reads_zero('cuda:0')
    assert calls == []
    state['capturing'] = False
    assert nat.lds_oob_reads_zero('cuda:0')
    state['capturing'] = True
    assert nat.lds_oob_reads_zero('cuda:0')          # cached: no launch, no refusal
    assert calls == ['cuda:0']
