"""CPU: the index algebra of the grid recurrence (lipreading_amd/csrc/lr_rnn_grid_map.h — fragment packing, MFMA lane
layout, publish / gather items of lr_rnn_grid.hip) played through on the host by oracle/grid_map_check.cpp and compared
with the plain products W_hh h and W_hh^T dG (nn.LSTM's recurrent half, better_model.py:47-49): exact."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
  exe = str(tmp_path_factory.mktemp("grid_map") / "grid_map_check")
  subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "oracle", "grid_map_check.cpp")], check=True)
  return exe


@pytest.mark.parametrize("H", [1536, 1400, 1156])
def test_grid_recurrence_layout_reproduces_the_matrix_products(checker, H):
  res = subprocess.run([checker, str(H)], capture_output=True, text=True, timeout=300)
  assert res.returncode == 0, res.stdout + res.stderr
  assert "forward max |err| 0, backward max |err| 0" in res.stdout, res.stdout
