"""lipreading_amd — MI355X-native hot path of joseph-zhong/LipReading.

landmark step -> recurrent sequence encoder -> CTC loss / greedy decode, as hand-written
gfx950 HIP kernels behind the C ABI in include/lipreading_hip.h, with a Python host side that
mirrors the reference's operator interface (VideoEncoder, ctc_loss, GreedyDecoder, train/eval).
"""
__version__ = "0.1.0"
