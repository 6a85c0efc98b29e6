// ppp_kernels.hpp -- launchers implemented by the .hip translation units.
#pragma once
#include "ppp_common.hpp"

namespace ppp {

hipError_t launch_consensus(const void *pred, int dtype, const uint8_t *ov, float *cons,
                            float *cnt, const Geo &G, hipStream_t s);
hipError_t launch_rank(const void *pred, int dtype, const float *cons, const uint8_t *ov,
                       float *score, const ppp_box &sb, const Geo &G, hipStream_t s);
hipError_t launch_patch_graph(const void *pred, int dtype, const float *cons,
                              const uint32_t *pairs, uint64_t n, float *aff, const Geo &G,
                              hipStream_t s);
hipError_t launch_label(const uint32_t *pairs, const float *aff, uint64_t n, uint32_t *cc_key,
                        void *work, const Geo &G, hipStream_t s);
hipError_t launch_paint(const void *pred, int dtype, const uint32_t *nodes,
                        const uint32_t *labels, uint64_t n, uint32_t *inst, const Geo &G,
                        hipStream_t s);
hipError_t launch_cons_to_reference(const float *compact, float *ref, const Geo &G,
                                    hipStream_t s);
hipError_t launch_patch_bits(const void *pred, int dtype, const uint32_t *centres, uint64_t n,
                             float thresh, uint32_t *bits, const Geo &G, hipStream_t s);
hipError_t launch_synth(const int32_t *labels, void *pred, int dtype, uint32_t seed, float hi,
                        float lo, float noise, const Geo &G, hipStream_t s);

}  // namespace ppp
