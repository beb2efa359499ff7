// Entry points declared in include/mqslam.h whose kernels have not landed yet.
// Each returns MQS_E_ARG with a clear message; this file shrinks as they are implemented.
#include "mqs_common.h"
#define MQS_TODO(name) do { mqs_set_error(name ": not implemented yet"); return MQS_E_ARG; } while (0)
extern "C" {
int mqs_match_knn2_f32(mqs_ctx *, const float *, int64_t, const float *, int64_t, int, int32_t *, float *) { MQS_TODO("mqs_match_knn2_f32"); }
int mqs_match_knn2_f32_dev(const float *, int64_t, const float *, int64_t, int, int32_t *, float *, void *) { MQS_TODO("mqs_match_knn2_f32_dev"); }
int mqs_match_knn2_f16(mqs_ctx *, const uint16_t *, int64_t, const uint16_t *, int64_t, int, int32_t *, float *) { MQS_TODO("mqs_match_knn2_f16"); }
int mqs_match_knn2_f16_dev(const uint16_t *, int64_t, const uint16_t *, int64_t, int, int32_t *, float *, void *, int64_t, void *) { MQS_TODO("mqs_match_knn2_f16_dev"); }
int64_t mqs_match_knn2_f16_workspace_bytes(int64_t, int64_t) { return 0; }
}
