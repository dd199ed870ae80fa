import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def networks():
    import json
    with open(os.path.join(ROOT, 'tests', 'golden', 'networks.json')) as fh:
        return json.load(fh)


def pytest_terminal_summary(terminalreporter):
    """UDS_TOL_REPORT=1: the largest observed / allowed error ratios of the session's close() calls."""
    if not os.environ.get('UDS_TOL_REPORT'):
        return
    from tests.util import OBSERVED
    worst = {}
    for name, line, err, lim in OBSERVED:
        key = (name, line)
        if key not in worst or err / lim > worst[key][0] / worst[key][1]:
            worst[key] = (err, lim)
    terminalreporter.write_line('observed / allowed error per close() call (UDS_TOL_REPORT):')
    for (name, line), (err, lim) in sorted(worst.items(), key=lambda kv: -kv[1][0] / kv[1][1]):
        terminalreporter.write_line('  %-110s line %4d  err %.2e  allowed %.2e  ratio %.3f' % (name, line, err, lim, err / lim))
