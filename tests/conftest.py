import json
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


@pytest.fixture(scope='session')
def goldens():
    with open(os.path.join(GOLDEN, 'goldens.json')) as fh:
        return json.load(fh)


def load_case(case):
    d = np.load(os.path.join(GOLDEN, case + '.npz'))
    return {k: d[k] for k in d.files}


@pytest.fixture(scope='session')
def spcfw():
    return load_case('q-SPC-FW')


@pytest.fixture(scope='session')
def heaq():
    return load_case('hydroxyethylaminoanthraquinone-in-water')


@pytest.fixture(scope='session')
def emim():
    return load_case('emim_BCN4_Jiung2014')


@pytest.fixture(scope='session')
def phenol():
    return load_case('phenol-in-water')
