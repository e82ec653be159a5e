// Stand-ins for what hulk_ingest.hip takes from the rest of libhulkhip.so and from the HIP runtime, so that the ingest path alone
// (hulk_parse_files: readers, gzip readers, block reader, parser; no GPU sink) links into a ThreadSanitizer build: tools/tsan_ingest.sh
#include <hip/hip_runtime.h>
#include "include/hulk_hip.h"
#include "hulk_amd/csrc/hulk_internal.h"
#include "hulk_amd/csrc/hulk_fastq.h"
namespace hulk {
// (the device FASTQ path of hulk_sketch_files is not run here — hulk_parse_files is host only — but it has to link)
int ctx_device(const hulk_ctx *) { return 0; }
int ctx_wait_event(hulk_ctx *, hipEvent_t) { return -1; }
int ctx_record_busy(hulk_ctx *, hipEvent_t, hipEvent_t, bool *) { return -1; }
void ctx_hint_host_offsets(hulk_ctx *, const uint64_t *) {}
hipError_t launch_fq_parse(hipStream_t, const FqBuffers &, const uint8_t *, const FqState *, uint8_t *, FqState *, uint32_t, uint64_t *, uint8_t *) { return hipErrorUnknown; }
hipError_t launch_fa_index(hipStream_t, const FaBuffers &, const uint8_t *, const FaState *, uint8_t *, FaState *, uint32_t) { return hipErrorUnknown; }
hipError_t launch_fa_place(hipStream_t, const FaBuffers &, const uint8_t *, const FaState *, uint8_t *, uint64_t, uint64_t *) { return hipErrorUnknown; }
hipStream_t ctx_stream(hulk_ctx *) { return nullptr; }
uint64_t ctx_min_read_len(const hulk_ctx *) { return 0; }
int ctx_fail(hulk_ctx *, int code, const char *) { return code; }
int ctx_stage_acquire(hulk_ctx *, size_t, uint64_t, StageSet *) { return -1; }
int ctx_stage_release(hulk_ctx *) { return 0; }
}
extern "C" {
const char *hipGetErrorString(hipError_t) { return "stub"; }
hipError_t hipMemcpyAsync(void *, const void *, size_t, hipMemcpyKind, hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *, unsigned) { return hipErrorUnknown; }
hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipFree(void *) { return hipSuccess; }
hipError_t hipHostFree(void *) { return hipSuccess; }
hipError_t hipHostMalloc(void **, size_t, unsigned) { return hipErrorUnknown; }
hipError_t hipMalloc(void **, size_t) { return hipErrorUnknown; }
hipError_t hipMemcpy(void *, const void *, size_t, hipMemcpyKind) { return hipErrorUnknown; }
hipError_t hipMemset(void *, int, size_t) { return hipErrorUnknown; }
hipError_t hipMemsetAsync(void *, int, size_t, hipStream_t) { return hipErrorUnknown; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *, unsigned) { return hipErrorUnknown; }
hipError_t hipDeviceGetStreamPriorityRange(int *, int *) { return hipErrorUnknown; }
hipError_t hipStreamCreateWithPriority(hipStream_t *, unsigned, int) { return hipErrorUnknown; }
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
int hulk_add_reads_device(hulk_ctx *, const uint8_t *, const uint64_t *, uint64_t, uint32_t, uint64_t) { return 0; }
const char *hulk_last_error(const hulk_ctx *) { return ""; }
const char *hulk_strerror(int) { return "error"; }
}
