"""Compile the gfx950 C-ABI library in-tree (lipreading_amd/_lib/liblipreading_hip.so).

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container as well
as on the MI355X box.  The built .so is git-ignored but travels with the repo snapshot.
"""
import glob
import hashlib
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "_lib")
LIB_PATH = os.path.join(LIB_DIR, "liblipreading_hip.so")
_STAMP = os.path.join(LIB_DIR, "liblipreading_hip.stamp")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "lipreading_hip.h")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-Wall", "-Wno-unused-function"]


def _hipcc():
  for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
    if cand and os.path.exists(cand):
      return cand
  raise RuntimeError("hipcc not found: the MI355X hot path cannot be built")


def sources():
  return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _fingerprint():
  h = hashlib.sha256()
  for p in sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [HEADER]:
    with open(p, "rb") as f:
      h.update(os.path.basename(p).encode())
      h.update(f.read())
  h.update(" ".join(HIPCC_FLAGS).encode())
  return h.hexdigest()


def is_current():
  if not (os.path.exists(LIB_PATH) and os.path.exists(_STAMP)):
    return False
  with open(_STAMP) as f:
    return f.read().strip() == _fingerprint()


def build_library(force=False, verbose=False):
  """Compile every .hip under csrc/ into one shared object.  Returns the .so path."""
  if not force and is_current():
    return LIB_PATH
  os.makedirs(LIB_DIR, exist_ok=True)
  cmd = [_hipcc()] + HIPCC_FLAGS + ["-o", LIB_PATH] + sources()
  if verbose:
    print(" ".join(cmd))
  res = subprocess.run(cmd, capture_output=True, text=True)
  if res.returncode != 0:
    raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
  if verbose and res.stderr:
    print(res.stderr)
  with open(_STAMP, "w") as f:
    f.write(_fingerprint())
  return LIB_PATH


if __name__ == "__main__":
  print(build_library(force=True, verbose=True))
