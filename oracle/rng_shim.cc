// TEST INFRASTRUCTURE ONLY (oracle): random sources for the CPU restatement.
//  * "std" source: std::mt19937 + std::uniform_real_distribution<double> / std::uniform_int_distribution<int>, i.e. the
//    same standard-library generator the reference wraps (src/mtrng/random.cc:29-47).  With it the oracle consumes the
//    same numbers as the reference for the same seed, which is how the oracle is pinned trajectory by trajectory.
//  * "philox" source: the product's counter-based stream (include/w2dyn_rng.h), used to check the GPU sampler.
#include <random>
#include <cstdint>
#include "cthyb_oracle.h"
#include "../include/w2dyn_rng.h"

namespace {
struct StdSource {
    std::mt19937 eng;
    std::uniform_real_distribution<double> d{0.0, 1.0};
};
double std_uniform(void *s) { auto *p = static_cast<StdSource *>(s); return p->d(p->eng); }
int std_randint(void *s, int lo, int hi) { auto *p = static_cast<StdSource *>(s); std::uniform_int_distribution<int> di(lo, hi); return di(p->eng); }
double px_uniform(void *s) { return w2dyn_uniform(static_cast<w2dyn_rng_state *>(s)); }
int px_randint(void *s, int lo, int hi) { return w2dyn_randint(static_cast<w2dyn_rng_state *>(s), lo, hi); }
}

extern "C" {
ora_rng *ora_rng_std(unsigned seed)
{
    auto *src = new StdSource(); src->eng.seed(seed);
    auto *r = new ora_rng{src, std_uniform, std_randint};
    return r;
}
ora_rng *ora_rng_philox(uint64_t seed, uint64_t walker)
{
    auto *st = new w2dyn_rng_state{seed, walker, 0};
    auto *r = new ora_rng{st, px_uniform, px_randint};
    return r;
}
uint64_t ora_rng_philox_ndraw(ora_rng *r) { return static_cast<w2dyn_rng_state *>(r->state)->ndraw; }
double ora_rng_uniform(ora_rng *r) { return r->uniform(r->state); }
int ora_rng_randint(ora_rng *r, int lo, int hi) { return r->randint(r->state, lo, hi); }
void ora_rng_free_std(ora_rng *r) { delete static_cast<StdSource *>(r->state); delete r; }
void ora_rng_free_philox(ora_rng *r) { delete static_cast<w2dyn_rng_state *>(r->state); delete r; }
}
