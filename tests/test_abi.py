"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/lipreading_hip.h declares (no compute calls without a GPU)."""
import os
import re

import pytest

from lipreading_amd import _build, _C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
  text = open(os.path.join(ROOT, "include", "lipreading_hip.h")).read()
  text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
  return sorted(set(re.findall(r"\b(lr_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
  path = _build.build_library()
  assert os.path.exists(path)
  lib = _C.lib()
  names = declared_symbols()
  assert len(names) >= 15
  for n in names:
    assert hasattr(lib, n), "header declares %s but the .so does not export it" % n
  # and the binding table covers the header exactly
  assert sorted(_C.SIGNATURES) == names


def test_version_and_status_strings():
  lib = _C.lib()
  assert lib.lr_version() >= 100
  assert lib.lr_status_string(0) == b"ok"
  assert b"workspace" in lib.lr_status_string(-2)
  assert lib.lr_ctc_workspace_bytes(32, 75, 65, 31) == (2 * 32 * 75 * 64 + 32 * 31 + 32 * 65) * 4
  # label length clamped to <= 256 (ctc_loss.py:46): 513 states -> stride 576
  assert lib.lr_ctc_workspace_bytes(1, 10, 65, 300) == (2 * 10 * 576 + 256 + 65) * 4


def test_null_arguments_are_rejected_without_a_device():
  lib = _C.lib()
  assert lib.lr_ctc_reduce(None, None, None, 1, None, None, None, 4, None) == -1
  assert lib.lr_lmk_translate(None, None, None, 1, 68, None) == -1


def test_product_path_has_no_cpu_fallback():
  import torch
  from lipreading_amd.ctc import ctc_loss_with_status
  lp = torch.zeros(2, 5, 65)
  with pytest.raises(_C.LipReadingHipError):
    ctc_loss_with_status(lp, torch.zeros(2, 3, dtype=torch.long), torch.tensor([5, 5]),
                         torch.tensor([3, 3]), "mean")


def test_product_package_never_imports_the_oracle():
  pkg = os.path.join(ROOT, "lipreading_amd")
  for dirpath, _, files in os.walk(pkg):
    for f in files:
      if f.endswith((".py", ".hip", ".h", ".cpp")):
        src = open(os.path.join(dirpath, f)).read()
        assert not re.search(r"^\s*(from|import)\s+oracle\b|import_module\(.oracle|oracle/_ref|lr_oracle",
                             src, flags=re.M), "%s reaches into oracle/" % f


def test_per_unit_compile_flags_name_existing_units_and_reach_the_tools():
  """_build.UNIT_FLAGS (flags of single translation units) must name sources that exist — a typo would silently build
  the unit without them — and the two shell tools that compile a unit on their own (variant libraries for A/B timing,
  the register-usage report) must repeat them, or they would time / report another kernel than the library holds."""
  units = {os.path.splitext(os.path.basename(p))[0] for p in _build.sources()}
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  for unit, flags in _build.UNIT_FLAGS.items():
    assert unit in units, unit
    assert flags and all(isinstance(f, str) for f in flags)
    for tool in ("tools/build_variant.sh", "tools/hip_resources.sh"):
      with open(os.path.join(root, tool)) as f:
        text = f.read()
      assert unit in text and " ".join(flags) in text, (tool, unit)
