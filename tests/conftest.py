import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


ONCHIP_LOG = os.path.join(ROOT, 'gpurun_out', 'onchip_launches_by_test.log')


@pytest.fixture(autouse=True)
def _onchip_launch_log(request):
    """GPU tests: the library appends a line per register-resident persistent launch (RRI_ONCHIP_LOG, rri_hip.hip); the
    summary at the end of the session says which tests -- the reference's golden vectors among them -- ran through that
    kernel rather than the launch-per-phase schedule"""
    if request.node.get_closest_marker('gpu') is None:
        yield
        return
    try:                                   # a record, not a requirement: no writable place, no record
        os.makedirs(os.path.dirname(ONCHIP_LOG), exist_ok=True)
        tmp = ONCHIP_LOG + '.current'
        if os.path.exists(tmp):
            os.remove(tmp)
        open(tmp, 'a').close()
        os.remove(tmp)
    except OSError:
        yield
        return
    old = os.environ.get('RRI_ONCHIP_LOG')
    os.environ['RRI_ONCHIP_LOG'] = tmp
    yield
    if old is None:
        os.environ.pop('RRI_ONCHIP_LOG', None)
    else:
        os.environ['RRI_ONCHIP_LOG'] = old
    try:
        if os.path.exists(tmp):
            lines = open(tmp).read().splitlines()
            os.remove(tmp)
            kinds = sorted(set(' '.join(ln.split()[3:5]) for ln in lines))
            with open(ONCHIP_LOG, 'a') as f:
                f.write('%-110s %4d persistent launches (%s)\n' % (request.node.nodeid, len(lines), ', '.join(kinds)))
    except OSError:
        pass


def pytest_sessionstart(session):
    try:
        if os.path.exists(ONCHIP_LOG):
            os.remove(ONCHIP_LOG)
    except OSError:
        pass


def pytest_terminal_summary(terminalreporter):
    if os.path.exists(ONCHIP_LOG):
        rows = open(ONCHIP_LOG).read().splitlines()
        total = sum(int(r.rsplit(' persistent launches', 1)[0].split()[-1]) for r in rows)
        terminalreporter.write_line('register-resident persistent sweep: %d launches in %d tests (%s)' % (total, len(rows), ONCHIP_LOG))


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


@pytest.fixture(scope='session')
def golden():
    return load_golden


def relfro(a, b):
    """relative Frobenius distance of a from the reference b"""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (den if den > 0 else 1.0)
