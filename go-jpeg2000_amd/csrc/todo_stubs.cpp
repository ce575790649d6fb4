// Temporary: entry points whose kernels are not written yet fail loudly.
#include "j2k_internal.h"
namespace j2k {
hipError_t launch_dwt97_fwd(hipStream_t, const LevelLaunch &, const void *, int, int32_t *, double *, double *, int, int, double, int) { return hipErrorNotSupported; }
hipError_t launch_dwt97_inv(hipStream_t, const LevelLaunch &, const void *, int, const double *, void *, int, int, int, int) { return hipErrorNotSupported; }
hipError_t launch_ht_encode(hipStream_t, const BlockJob *, int, const int32_t *, uint8_t *, uint32_t *, uint8_t *, int *) { return hipErrorNotSupported; }
hipError_t launch_ht_decode(hipStream_t, const BlockJob *, int, const uint8_t *, const uint64_t *, const uint32_t *, int32_t *) { return hipErrorNotSupported; }
hipError_t launch_t1_encode(hipStream_t, const BlockJob *, int, const int32_t *, uint8_t *, uint32_t *, uint8_t *, uint8_t *, size_t, int *) { return hipErrorNotSupported; }
hipError_t launch_t1_decode(hipStream_t, const BlockJob *, int, const uint8_t *, const uint64_t *, const uint32_t *, const uint8_t *, int32_t *, uint8_t *, size_t) { return hipErrorNotSupported; }
size_t t1_work_bytes(int, int) { return 0; }
hipError_t launch_compact(hipStream_t, const BlockJob *, int, const uint8_t *, const uint32_t *, uint64_t *, uint8_t *, void *) { return hipErrorNotSupported; }
}
