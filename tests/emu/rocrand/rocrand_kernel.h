// TEST INFRASTRUCTURE ONLY -- stand-in for rocRAND's device API under the CPU emulator.
// Philox4x32-10 counter-based generator + Box-Muller normals.  It follows the published
// Philox algorithm (Salmon et al., SC'11) but makes no attempt to reproduce rocRAND's
// exact counter/offset conventions: emulator tests only check distribution and stream
// independence, never specific values.
#pragma once
#include <hip/hip_runtime.h>

struct rocrand_state_philox4x32_10 {
  uint32_t ctr[4];
  uint32_t key[2];
  uint32_t out[4];
  int have;
};

static inline void emu_philox_round(uint32_t* c, const uint32_t* k) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
  const uint32_t n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
  const uint32_t n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

static inline void emu_philox_gen(rocrand_state_philox4x32_10* s) {
  uint32_t c[4] = {s->ctr[0], s->ctr[1], s->ctr[2], s->ctr[3]};
  uint32_t k[2] = {s->key[0], s->key[1]};
  for (int r = 0; r < 10; ++r) {
    emu_philox_round(c, k);
    k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
  }
  for (int i = 0; i < 4; ++i) s->out[i] = c[i];
  if (++s->ctr[0] == 0) if (++s->ctr[1] == 0) if (++s->ctr[2] == 0) ++s->ctr[3];
}

static inline void rocrand_init(unsigned long long seed, unsigned long long subsequence, unsigned long long offset,
                                rocrand_state_philox4x32_10* s) {
  s->key[0] = (uint32_t)seed; s->key[1] = (uint32_t)(seed >> 32);
  s->ctr[0] = (uint32_t)(offset >> 2); s->ctr[1] = (uint32_t)(offset >> 34);
  s->ctr[2] = (uint32_t)subsequence; s->ctr[3] = (uint32_t)(subsequence >> 32);
  s->have = 0;
}

static inline float4 rocrand_normal4(rocrand_state_philox4x32_10* s) {
  emu_philox_gen(s);
  float u[4];
  for (int i = 0; i < 4; ++i) u[i] = ((float)s->out[i] + 0.5f) * 2.3283064365386963e-10f;   // (0, 1)
  const float r0 = sqrtf(-2.0f * logf(u[0])), r1 = sqrtf(-2.0f * logf(u[2]));
  const float t0 = 6.283185307179586f * u[1], t1 = 6.283185307179586f * u[3];
  return make_float4(r0 * cosf(t0), r0 * sinf(t0), r1 * cosf(t1), r1 * sinf(t1));
}

static inline float rocrand_normal(rocrand_state_philox4x32_10* s) { return rocrand_normal4(s).x; }
static inline uint4 rocrand4(rocrand_state_philox4x32_10* s) {
  emu_philox_gen(s);
  return make_uint4(s->out[0], s->out[1], s->out[2], s->out[3]);
}
