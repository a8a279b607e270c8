"""Compile the gfx950 C-ABI library in-tree (lipreading_amd/_lib/liblipreading_hip.so).

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container as well
as on the MI355X box.  The built .so is git-ignored but travels with the repo snapshot.

Every csrc/*.hip becomes its own object file (compiled in parallel, re-used while the source,
the shared headers and the flags are unchanged) and the objects are linked into one shared
object: editing one kernel file costs one compile, not the whole library.
"""
import glob
import hashlib
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "_lib")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB_PATH = os.path.join(LIB_DIR, "liblipreading_hip.so")
_STAMP = os.path.join(LIB_DIR, "liblipreading_hip.stamp")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "lipreading_hip.h")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
               "-Wall", "-Wno-unused-function"]
# per translation unit (tools/build_variant.sh repeats them).  lr_conv1: MFMA results in VGPRs — the first layer's
# kernels have registers to spare and are bound by their vector instructions; the accumulators' way through AGPRs
# costs the forward 32 v_accvgpr_read per tile.
UNIT_FLAGS = {"lr_conv1": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _flags(src):
  return HIPCC_FLAGS + UNIT_FLAGS.get(os.path.splitext(os.path.basename(src))[0], [])


def _hipcc():
  for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
    if cand and os.path.exists(cand):
      return cand
  raise RuntimeError("hipcc not found: the MI355X hot path cannot be built")


def sources():
  return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _shared_inputs():
  return sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [HEADER]


def _digest(paths, extra=""):
  h = hashlib.sha256()
  for p in paths:
    with open(p, "rb") as f:
      h.update(os.path.basename(p).encode())
      h.update(f.read())
  h.update(extra.encode())
  return h.hexdigest()


def _fingerprint():
  return _digest(sources() + _shared_inputs(), " ".join(HIPCC_FLAGS) + repr(sorted(UNIT_FLAGS.items())))


def is_current():
  if not (os.path.exists(LIB_PATH) and os.path.exists(_STAMP)):
    return False
  with open(_STAMP) as f:
    return f.read().strip() == _fingerprint()


def _compile_one(src, verbose):
  """src -> obj (skipped when the object was built from the same inputs).  Returns the object path."""
  name = os.path.splitext(os.path.basename(src))[0]
  obj = os.path.join(OBJ_DIR, name + ".o")
  stamp = obj + ".stamp"
  want = _digest([src] + _shared_inputs(), " ".join(_flags(src)))
  if os.path.exists(obj) and os.path.exists(stamp):
    with open(stamp) as f:
      if f.read().strip() == want:
        return obj
  cmd = [_hipcc()] + _flags(src) + ["-c", src, "-o", obj]
  if verbose:
    print(" ".join(cmd), flush=True)
  res = subprocess.run(cmd, capture_output=True, text=True)
  if res.returncode != 0:
    raise RuntimeError("hipcc failed on %s:\n%s%s" % (src, res.stdout, res.stderr))
  if verbose and res.stderr:
    print(res.stderr)
  with open(stamp, "w") as f:
    f.write(want)
  return obj


def build_library(force=False, verbose=False):
  """Compile every .hip under csrc/ and link one shared object.  Returns the .so path."""
  if not force and is_current():
    return LIB_PATH
  os.makedirs(OBJ_DIR, exist_ok=True)
  if force:
    for p in glob.glob(os.path.join(OBJ_DIR, "*.stamp")):
      os.remove(p)
  srcs = sources()
  with ThreadPoolExecutor(max_workers=min(len(srcs), max(1, (os.cpu_count() or 2)))) as pool:
    objs = list(pool.map(lambda s: _compile_one(s, verbose), srcs))
  cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
  if verbose:
    print(" ".join(cmd), flush=True)
  res = subprocess.run(cmd, capture_output=True, text=True)
  if res.returncode != 0:
    raise RuntimeError("link failed:\n" + res.stdout + res.stderr)
  with open(_STAMP, "w") as f:
    f.write(_fingerprint())
  return LIB_PATH


if __name__ == "__main__":
  print(build_library(force=True, verbose=True))
