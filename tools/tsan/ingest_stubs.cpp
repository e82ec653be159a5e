// Stand-ins for what hulk_ingest.hip takes from the rest of libhulkhip.so and from the HIP runtime, so that the ingest path alone
// (hulk_parse_files: readers, gzip readers, block reader, parser; no GPU sink) links into a ThreadSanitizer build: tools/tsan_ingest.sh
#include <hip/hip_runtime.h>
#include "include/hulk_hip.h"
#include "hulk_amd/csrc/hulk_internal.h"
namespace hulk {
hipStream_t ctx_stream(hulk_ctx *) { return nullptr; }
uint64_t ctx_min_read_len(const hulk_ctx *) { return 0; }
int ctx_fail(hulk_ctx *, int code, const char *) { return code; }
int ctx_stage_acquire(hulk_ctx *, size_t, uint64_t, StageSet *) { return -1; }
int ctx_stage_release(hulk_ctx *) { return 0; }
}
extern "C" {
const char *hipGetErrorString(hipError_t) { return "stub"; }
hipError_t hipMemcpyAsync(void *, const void *, size_t, hipMemcpyKind, hipStream_t) { return hipSuccess; }
int hulk_add_reads_device(hulk_ctx *, const uint8_t *, const uint64_t *, uint64_t, uint32_t, uint64_t) { return 0; }
const char *hulk_last_error(const hulk_ctx *) { return ""; }
const char *hulk_strerror(int) { return "error"; }
}
