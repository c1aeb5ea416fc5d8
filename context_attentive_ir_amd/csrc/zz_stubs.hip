// TEMPORARY: entry points not implemented yet report NIR_ERR_UNSUPPORTED.
#include "common.hpp"
extern "C" size_t nir_duet_workspace_bytes(int, int, int, int, int, const nir_duet_weights*) { return 0; }
extern "C" int nir_duet_score(const int64_t*, const int64_t*, int, int, int, int, const float*, int64_t, int,
                              const nir_duet_weights*, void*, size_t, float*, float*, float*, nir_stream_t) {
    nir::set_error("duet: not implemented yet"); return NIR_ERR_UNSUPPORTED; }
extern "C" size_t nir_cars_encode_workspace_bytes(int64_t, int, int, const nir_cars_encoder_weights*) { return 0; }
extern "C" int nir_cars_encode(const int64_t*, const int64_t*, int64_t, int, const float*, int64_t, int,
                               const nir_cars_encoder_weights*, void*, size_t, float*, float*, nir_stream_t) {
    nir::set_error("cars_encode: not implemented yet"); return NIR_ERR_UNSUPPORTED; }
extern "C" size_t nir_cars_session_workspace_bytes(int, int, int, const nir_cars_session_weights*) { return 0; }
extern "C" int nir_cars_rank_session(const float*, const float*, const float*, int, int, int,
                                     const nir_cars_session_weights*, void*, size_t, float*, float*, nir_stream_t) {
    nir::set_error("cars_rank_session: not implemented yet"); return NIR_ERR_UNSUPPORTED; }
