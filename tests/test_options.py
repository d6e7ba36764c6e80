"""DLWP.options: the one environment variable of the engine's switchable behaviours."""
import os

import pytest


def test_options_defaults_and_overrides():
    from DLWP.options import DEFAULTS, option
    os.environ.pop('DLWPCS_OPTIONS', None)
    for k, v in DEFAULTS.items():
        assert option(k) is v
    os.environ['DLWPCS_OPTIONS'] = 'premask=0, check_finite=1,graphs=off'
    try:
        assert option('premask') is False and option('check_finite') is True and option('graphs') is False
        assert option('fuse_pool') is True
        with pytest.raises(KeyError):
            option('no_such_option')
        os.environ['DLWPCS_OPTIONS'] = 'no_such_option=1'
        with pytest.raises(KeyError):
            option('premask')
    finally:
        os.environ.pop('DLWPCS_OPTIONS', None)
