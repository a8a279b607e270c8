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
# step; with 8 queues it equals the single-process step).  Only read when the HIP runtime initialises: import this
# package (or set the variable yourself) before the first CUDA call of the process.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "6")
