"""lipreading_amd — MI355X-native hot path of joseph-zhong/LipReading.

landmark step -> recurrent sequence encoder -> CTC loss / greedy decode, as hand-written
gfx950 HIP kernels behind the C ABI in include/lipreading_hip.h, with a Python host side that
mirrors the reference's operator interface (VideoEncoder, ctc_loss, GreedyDecoder, train/eval).
"""
__version__ = "0.1.0"

import os as _os

# HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), round robin.  A data-parallel step
# runs on the main stream, the encoder's and the conv frontend's weight-gradient side streams, GradSync's collective
# stream and RCCL's own: with four queues two of them share one, and what was meant to overlap runs back to back
# (round 5: the 1-rank RCCL step lost its encoder side stream to the main stream's queue, +85 us of a 2.57 ms pixel
# step).  Six queues fix that — for a process that is a RANK.  They are NOT a better default for everybody: a
# single-process pixel step REPLAYED AS A hipGraph takes 3.4-3.5 ms with six queues where it takes 2.55 with four, five
# or eight (eager launches: 2.43-2.47 with any of them; measured in the same visit, profiles/r05_variants_ab.txt), and
# with eight the graph-replayed landmark step of a rank is 2.7x slower.  So: six when the process was launched as one
# of several ranks (WORLD_SIZE > 1, what torch.distributed.run sets) or as bench.py's one-rank stand-in for that path,
# HIP's default otherwise; a launcher that spawns its ranks some other way sets GPU_MAX_HW_QUEUES=6 itself.  Only read
# when the HIP runtime initialises: import this package (or set the variable) before the first CUDA call of the process.
def _is_a_rank():
  try:
    return int(_os.environ.get("WORLD_SIZE", "1")) > 1 or _os.environ.get("LIPREADING_BENCH_FORCE_DIST", "0") == "1"
  except ValueError:
    return False


if _is_a_rank():
  _os.environ.setdefault("GPU_MAX_HW_QUEUES", "6")
