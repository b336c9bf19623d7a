/*
 * npb_seeds.h -- per-plant random streams of the data-generation scenarios, for whole arrays of seeds (host side).
 *
 * The reference randomises one plant at a time: get_randomized_feedwater_conditions(action, seed)
 * (data_gen/config_engine/initial_conditions/randomization_utils.py:799-842) seeds BOTH generators with the scenario seed --
 * `random.seed(seed)` and `np.random.seed(seed)` (:812-813) -- then draws the weighted scenario pick and the uniform parameters
 * from the stdlib generator (`random.random()` :857, `random.uniform()` :876,886) and the normal parameters from numpy's
 * legacy global generator (`np.random.normal`, :881); add_randomness_to_conditions (:13-122) draws one
 * `np.random.uniform` per numeric leaf (:80,93,108) after `np.random.seed(seed)` (:33).  For 10^5-10^6 plants the generator SET-UP dominates
 * (a Mersenne Twister is seeded per plant: 1 247 dependent steps for CPython, 623 for numpy), so these entry points do it for
 * an array of seeds at once, sixteen seeds per SIMD block, blocks over host threads.
 *
 * Both generators are third-party code the reference depends on, not part of /root/reference; what is restated is their
 * published algorithm (MT19937, Matsumoto & Nishimura 1998, and the two projects' documented seeding / output rules):
 *   CPython 3.10  Modules/_randommodule.c: random_seed (int -> abs -> 32-bit digits, least significant first -> init_by_array),
 *                 random_random ((a >> 5) * 2^26 + (b >> 6)) / 2^53
 *   numpy 2.2     numpy/random/_legacy seeding (_mt19937.pyx _legacy_seeding: an int in [0, 2^32) -> init_genrand),
 *                 random_standard_uniform (the same 53-bit rule), legacy_gauss (legacy-distributions.c: polar Box-Muller with a
 *                 one-value cache)
 * pinned by tests/test_seeds_cpu.py against the interpreter's own `random.Random` and `numpy.random.RandomState`.
 *
 * out is row-major [n][k]; seeds are non-negative (np: below 2^32, as numpy itself requires).  Return 0, or NPB_EINVAL.
 */
#ifndef NPB_SEEDS_H
#define NPB_SEEDS_H
#include <stddef.h>
#include <stdint.h>
#ifndef NPB_API
#define NPB_API __attribute__((visibility("default")))
#endif
#ifdef __cplusplus
extern "C" {
#endif
/* random.Random(seed): the first k values of .random() */
NPB_API int npb_seed_py_random(const int64_t *seeds, size_t n, int k, double *out);
/* numpy.random.RandomState(seed): the first k values of .random_sample() (what .uniform(lo, hi) scales: lo + (hi - lo) * u) */
NPB_API int npb_seed_np_random(const int64_t *seeds, size_t n, int k, double *out);
/* numpy.random.RandomState(seed): the first k values of .standard_normal() (what .normal(loc, scale) scales: loc + scale * g) */
NPB_API int npb_seed_np_gauss(const int64_t *seeds, size_t n, int k, double *out);
/* host threads the three calls above use (0 = as many as the process may run on, at most 16) */
NPB_API int npb_seed_set_threads(int threads);
#ifdef __cplusplus
}
#endif
#endif
