// mc_rng.h -- the counter-based random numbers of the Monte-Carlo front-ends (mc_frontend.hip) and of
// the host-side inspection entry points that must name the same frames (wrappers.cpp).
// Every (seed, global frame index, stream) names a splitmix64 sequence: draw i = mix64(key + (i+1) GOLDEN),
// so a frame's data does not depend on batch size, GPU count or launch order.
#pragma once
#include <cstdint>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define LDPC_HD __host__ __device__
#else
#define LDPC_HD
#endif

namespace ldpc_rng {
constexpr uint64_t GOLDEN = 0x9E3779B97F4A7C15ull;
LDPC_HD inline uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// stream: 0 source symbols, 1 channel
LDPC_HD inline uint64_t stream_key(uint64_t seed, uint64_t frame, uint64_t stream) {
  return mix64(seed ^ mix64(frame * 4 + stream + GOLDEN));
}
LDPC_HD inline uint64_t draw(uint64_t key, uint64_t i) { return mix64(key + (i + 1) * GOLDEN); }
// uniform integer below q from the high half of a draw
LDPC_HD inline uint32_t below(uint64_t r, uint32_t q) { return (uint32_t)(((r >> 32) * (uint64_t)q) >> 32); }
// the percent-granular uniform of GFq_LDPC.c channel() (:218, rand() * 100 / RAND_MAX as floats), from 24 bits of a draw
LDPC_HD inline float percent(uint64_t r) { return (float)(r >> 40) * 100.0f / (float)((1u << 24) - 1); }
}  // namespace ldpc_rng
