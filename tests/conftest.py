import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'dlwp-cs_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu on the GPU box)')
    # A process that dies on a GPU box (HIP runtime abort, memory fault) would write a core of its whole address space -- with
    # the device's 288 GB mapped that takes minutes and looks like a hang.  No cores; the python stacks of a fatal
    # signal go to stderr instead.  Spawned workers inherit both (the limit through the process, the handler through the env).
    try:
        import resource
        resource.setrlimit(resource.RLIMIT_CORE, (0, 0))
    except (ImportError, ValueError, OSError):
        pass
    import faulthandler
    faulthandler.enable()
    os.environ.setdefault('PYTHONFAULTHANDLER', '1')


def pytest_collection_modifyitems(config, items):
    # a GPU test that hangs (a spawned rank that died while its peer waits in a collective) fails after 10 minutes instead of
    # taking the whole suite's time limit with it (pytest-timeout, when the image has it)
    if not config.pluginmanager.hasplugin('timeout'):
        return
    for it in items:
        if it.get_closest_marker('gpu') is not None and it.get_closest_marker('timeout') is None:
            it.add_marker(pytest.mark.timeout(600))


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
