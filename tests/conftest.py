import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


SSY_ARR = ["h_lam_states", "h_lam_Q", "h_c_states", "h_c_Q", "h_z_states", "h_z_Q",
           "z_states", "z_Q", "sigma_c_states", "sigma_z_states"]
GCY_ARR = ["z_states", "z_Q", "z_pi_states", "z_pi_Q", "h_z_states", "h_z_Q", "sigma_z_states",
           "h_c_states", "h_c_Q", "sigma_c_states", "h_zpi_states", "h_zpi_Q", "sigma_zpi_states",
           "h_lam_states", "h_lam_Q"]


def golden_arrays(g, model):
    names = SSY_ARR if model == "ssy" else GCY_ARR
    return tuple(g["arr_" + n] for n in names)


@pytest.fixture(scope="session")
def have_gpu():
    import torch
    return torch.cuda.is_available()
