// model.hip -- placeholder until the Inception-v3 kernels land.
#include "dv_internal.h"

struct dv_model { int unused; };

extern "C" {
int dv_model_create(const dv_model_desc*, int, dv_model**) {
  return dv::fail(DV_ERR_UNSUPPORTED, "dv_model: not built yet");
}
void dv_model_destroy(dv_model*) {}
int64_t dv_model_num_params(const dv_model*) { return 0; }
int dv_model_num_layers(const dv_model*) { return 0; }
int dv_model_layer_info(const dv_model*, int, int32_t*, int32_t*, int32_t*, int32_t*, int64_t*) {
  return dv::fail(DV_ERR_UNSUPPORTED, "dv_model: not built yet");
}
int dv_model_load_weights(dv_model*, const float*, int64_t) {
  return dv::fail(DV_ERR_UNSUPPORTED, "dv_model: not built yet");
}
int dv_model_infer(dv_model*, const uint8_t*, int, float*, void*) {
  return dv::fail(DV_ERR_UNSUPPORTED, "dv_model: not built yet");
}
}
