// temporary: entry points not implemented yet report LR_ERR_UNSUPPORTED (removed as they land)
#include "lr_common.h"
extern "C" int lr_sgemm(int, int, int, int, int, float, const float*, int, const float*, int, float,
                        float*, int, const float*, int, int, lr_stream_t) { return LR_ERR_UNSUPPORTED; }
extern "C" size_t lr_rnn_reserve_bytes(int, int, int, int, int, int) { return 0; }
extern "C" size_t lr_rnn_workspace_bytes(int, int, int, int, int, int) { return 0; }
extern "C" int lr_rnn_layer_forward(int, const float*, const int32_t*, const float* const*,
                                    const float* const*, const float* const*, const float* const*,
                                    float*, float*, float*, void*, size_t, int, int, int, int, int,
                                    lr_stream_t) { return LR_ERR_UNSUPPORTED; }
extern "C" int lr_rnn_layer_backward(int, const float*, const int32_t*, const float* const*,
                                     const float* const*, const float* const*, const float* const*,
                                     const float*, const float*, const float*, const float*, float*,
                                     float* const*, float* const*, float* const*, float* const*,
                                     const void*, size_t, void*, size_t, int, int, int, int, int,
                                     lr_stream_t) { return LR_ERR_UNSUPPORTED; }
extern "C" int lr_proj_logsoftmax_forward(const float*, const float*, const float*, const float*,
                                          float*, int, int, int, lr_stream_t) { return LR_ERR_UNSUPPORTED; }
extern "C" int lr_proj_logsoftmax_backward(const float*, const float*, const float*, const float*,
                                           float*, float*, float*, float*, int, int, int,
                                           lr_stream_t) { return LR_ERR_UNSUPPORTED; }
