import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
  config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


def load_golden(name):
  """Flat npz -> nested dict keyed by the '/'-separated path components."""
  z = np.load(os.path.join(GOLDEN, name), allow_pickle=True)
  out = {}
  for k in z.files:
    d = out
    parts = k.split("/")
    for p in parts[:-1]:
      d = d.setdefault(p, {})
    d[parts[-1]] = z[k]
  return out


@pytest.fixture(scope="session")
def golden_ctc():
  return load_golden("ctc_cases.npz")


@pytest.fixture(scope="session")
def golden_enc():
  return load_golden("enc_cases.npz")


@pytest.fixture(scope="session")
def golden_greedy():
  return load_golden("greedy_cases.npz")


@pytest.fixture(scope="session")
def golden_step():
  return load_golden("step_cases.npz")


@pytest.fixture(scope="session")
def golden_lmk():
  return load_golden("lmk_cases.npz")


@pytest.fixture(scope="session")
def golden_enc2():
  return load_golden("enc2_cases.npz")


@pytest.fixture(scope="session")
def golden_dec2():
  return load_golden("dec2_cases.npz")


@pytest.fixture(scope="session")
def golden_prn():
  return load_golden("prn_cases.npz")


def has_gpu():
  try:
    import torch
    return torch.cuda.is_available()
  except Exception:
    return False


@pytest.fixture(scope="session")
def golden_dec():
  return load_golden("dec_cases.npz")
