/* Counter-based random stream of one walker (Philox4x32-10, Salmon et al. SC'11).
 *
 * Replaces the reference's process-wide std::mt19937 singleton (src/mtrng/random.cc:29-47, Random.F90:36-57):
 * SURVEY.md §2 item 8 - bit parity with mt19937 is not a goal, reproducibility per (seed, walker) is.
 * The stream is a pure function of (seed, walker id, draw index), so a walker needs only a 64-bit draw counter.
 * Host (oracle, tests) and device (HIP kernels) include this same header, which is what makes trajectory-level
 * parity between the CPU oracle and the GPU sampler possible.
 */
#ifndef W2DYN_RNG_H
#define W2DYN_RNG_H
#include <stdint.h>

#if defined(__HIPCC__)
#define W2DYN_HD __host__ __device__ inline
#else
#define W2DYN_HD static inline
#endif

typedef struct w2dyn_rng_state {
    uint64_t seed;    /* NSeed of the run                      */
    uint64_t walker;  /* global walker (chain) number          */
    uint64_t ndraw;   /* number of uniforms consumed so far    */
} w2dyn_rng_state;

W2DYN_HD void w2dyn_philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

/* uniform double in [0,1) with 53 random bits; draw i uses half (i&1) of Philox block (i>>1) */
W2DYN_HD double w2dyn_uniform_at(uint64_t seed, uint64_t walker, uint64_t idx)
{
    uint64_t blk = idx >> 1;
    uint32_t c[4] = { (uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)walker, (uint32_t)(walker >> 32) };
    w2dyn_philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    uint32_t a = (idx & 1) ? c[2] : c[0];
    uint32_t b = (idx & 1) ? c[3] : c[1];
    uint64_t m = ((uint64_t)(a >> 5) << 26) | (uint64_t)(b >> 6);
    return (double)m * (1.0 / 9007199254740992.0);
}

W2DYN_HD double w2dyn_uniform(w2dyn_rng_state *s)
{
    double u = w2dyn_uniform_at(s->seed, s->walker, s->ndraw);
    s->ndraw += 1;
    return u;
}

/* integer in [lo, hi], one uniform per call */
W2DYN_HD int w2dyn_randint(w2dyn_rng_state *s, int lo, int hi)
{
    int n = hi - lo + 1;
    int r = (int)(w2dyn_uniform(s) * (double)n);
    if (r >= n) r = n - 1;
    return lo + r;
}

#endif
