/* sar_hip_debug.h -- diagnostic entry points of libsar_hip.so.  NOT part of the product ABI: they are compiled only into a
 * `make -C skeleton-action-recognition_amd/csrc DEBUG=1` build (-DSAR_DEBUG) and used by tools/kernel_repeat_check.py. */
#ifndef SAR_HIP_DEBUG_H
#define SAR_HIP_DEBUG_H
#include "sar_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* fills 64 KiB of LDS on every CU with `pattern` (a kernel whose result depends on LDS it never wrote becomes visible);
 * sink: 4 device bytes */
int sar_debug_poison_lds(unsigned pattern, void* sink, sar_stream_t s);
/* workgroups/CU the runtime predicts for the temporal GEMM at a dynamic-LDS size */
int sar_debug_occupancy(int which, int lds_bytes);
#ifdef __cplusplus
}
#endif
#endif
