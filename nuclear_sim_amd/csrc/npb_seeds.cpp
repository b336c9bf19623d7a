/*
 * npb_seeds.cpp -- include/npb_seeds.h: Mersenne-Twister streams for whole arrays of seeds (host code of libnpb.so).
 *
 * Sixteen seeds share a block: the state is kept word-major, mt[i][lane], so that the dependent chains of the seeding rules
 * (each word a function of the one before) run as sixteen independent chains in the lanes of one SIMD instruction stream,
 * and so do the twist and the tempering.  Blocks are distributed over host threads.
 */
#include <math.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <thread>
#include <vector>
#include "../../include/npb_seeds.h"

namespace {

constexpr int MT_N = 624, MT_M = 397, LANES = 16;
constexpr uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX_A = 0x9908b0dfu;

struct Block {
  uint32_t mt[MT_N][LANES];
  uint32_t out[MT_N][LANES];   /* tempered words of the current twist */
};

/* init_genrand(s): mt[0] = s, mt[i] = 1812433253 * (mt[i-1] ^ (mt[i-1] >> 30)) + i */
#define NPB_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))

NPB_CLONES void seed_genrand(Block &b, const uint32_t *s) {
  for (int l = 0; l < LANES; l++) b.mt[0][l] = s[l];
  for (int i = 1; i < MT_N; i++)
    for (int l = 0; l < LANES; l++) b.mt[i][l] = 1812433253u * (b.mt[i - 1][l] ^ (b.mt[i - 1][l] >> 30)) + (uint32_t)i;
}

/* init_by_array(key, len) with len = 1 or 2 per lane (CPython: the 32-bit digits of abs(seed), at least one) */
const uint32_t *genrand_19650218() {
  static uint32_t base[MT_N];
  static bool done = false;
  if (!done) {
    base[0] = 19650218u;
    for (int i = 1; i < MT_N; i++) base[i] = 1812433253u * (base[i - 1] ^ (base[i - 1] >> 30)) + (uint32_t)i;
    done = true;
  }
  return base;
}

NPB_CLONES void seed_by_array(Block &b, const uint32_t *key0, const uint32_t *key1, const uint32_t *len, const uint32_t *base) {
  for (int i = 0; i < MT_N; i++)
    for (int l = 0; l < LANES; l++) b.mt[i][l] = base[i];
  /* first loop: k = max(N, len) = N iterations; j cycles through the key (j = t mod len) */
  int i = 1;
  for (int t = 0; t < MT_N; t++) {
    const int prev = i - 1;
    for (int l = 0; l < LANES; l++) {
      const uint32_t j = len[l] == 2 ? (uint32_t)(t & 1) : 0u;
      const uint32_t kj = j ? key1[l] : key0[l];
      const uint32_t p = b.mt[prev][l];
      b.mt[i][l] = (b.mt[i][l] ^ ((p ^ (p >> 30)) * 1664525u)) + kj + j;
    }
    i++;
    if (i >= MT_N) {
      for (int l = 0; l < LANES; l++) b.mt[0][l] = b.mt[MT_N - 1][l];
      i = 1;
    }
  }
  for (int t = 0; t < MT_N - 1; t++) {
    const int prev = i - 1;
    for (int l = 0; l < LANES; l++) {
      const uint32_t p = b.mt[prev][l];
      b.mt[i][l] = (b.mt[i][l] ^ ((p ^ (p >> 30)) * 1566083941u)) - (uint32_t)i;
    }
    i++;
    if (i >= MT_N) {
      for (int l = 0; l < LANES; l++) b.mt[0][l] = b.mt[MT_N - 1][l];
      i = 1;
    }
  }
  for (int l = 0; l < LANES; l++) b.mt[0][l] = 0x80000000u;
}

/* one generation of 624 words: the twist in place, then the tempering into b.out */
NPB_CLONES void twist(Block &b) {
  for (int k = 0; k < MT_N; k++) {
    const int k1 = (k + 1) % MT_N, km = (k + MT_M) % MT_N;
    for (int l = 0; l < LANES; l++) {
      const uint32_t y = (b.mt[k][l] & UPPER) | (b.mt[k1][l] & LOWER);
      b.mt[k][l] = b.mt[km][l] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
    }
  }
  for (int k = 0; k < MT_N; k++)
    for (int l = 0; l < LANES; l++) {
      uint32_t y = b.mt[k][l];
      y ^= (y >> 11);
      y ^= (y << 7) & 0x9d2c5680u;
      y ^= (y << 15) & 0xefc60000u;
      y ^= (y >> 18);
      b.out[k][l] = y;
    }
}

inline double to_double(uint32_t a, uint32_t bb) { return ((double)(a >> 5) * 67108864.0 + (double)(bb >> 6)) / 9007199254740992.0; }

enum Kind { PY_RANDOM, NP_RANDOM, NP_GAUSS };

void run_blocks(Kind kind, const int64_t *seeds, size_t n, int k, double *out, size_t block0, size_t block1) {
  Block *b = nullptr;
  if (posix_memalign((void **)&b, 64, sizeof(Block)) != 0) return;
  const uint32_t *base = genrand_19650218();
  std::vector<uint32_t> words;     /* NP_GAUSS: the tempered words of every generation so far, [generation * 624 + word][lane] */
  for (size_t blk = block0; blk < block1; blk++) {
    const size_t first = blk * LANES;
    const int live = (int)std::min<size_t>(LANES, n - first);
    uint32_t k0[LANES], k1[LANES], len[LANES];
    for (int l = 0; l < LANES; l++) {
      const uint64_t s = (uint64_t)seeds[first + (l < live ? l : 0)];
      k0[l] = (uint32_t)s; k1[l] = (uint32_t)(s >> 32); len[l] = k1[l] ? 2u : 1u;
    }
    if (kind == PY_RANDOM) seed_by_array(*b, k0, k1, len, base); else seed_genrand(*b, k0);
    if (kind != NP_GAUSS) {
      int used = MT_N;       /* words of the current generation consumed */
      for (int j = 0; j < k; j++) {
        if (used + 2 > MT_N) {
          /* 624 is even and two words are taken at a time, so a pair never straddles a generation */
          twist(*b); used = 0;
        }
        for (int l = 0; l < live; l++) out[(first + l) * (size_t)k + j] = to_double(b->out[used][l], b->out[used + 1][l]);
        used += 2;
      }
    } else {
      /* legacy_gauss: x1, x2 = 2 u - 1 until 0 < r2 < 1; f = sqrt(-2 log r2 / r2); returns f * x2 and caches f * x1 for the
       * next call.  Lanes reject independently, so each walks the block's word stream with its own cursor; generations are
       * appended for the whole block when any lane runs past the end. */
      words.clear();
      int generations = 0;
      auto need = [&](int pos) {
        while (pos + 2 > generations * MT_N) {
          twist(*b);
          words.resize((size_t)(generations + 1) * MT_N * LANES);
          memcpy(&words[(size_t)generations * MT_N * LANES], b->out, sizeof(b->out));
          generations++;
        }
      };
      for (int l = 0; l < live; l++) {
        int pos = 0, j = 0;
        while (j < k) {
          double x1, x2, r2;
          do {
            need(pos + 2);   /* four words */
            const uint32_t a1 = words[(size_t)pos * LANES + l], b1 = words[(size_t)(pos + 1) * LANES + l];
            const uint32_t a2 = words[(size_t)(pos + 2) * LANES + l], b2 = words[(size_t)(pos + 3) * LANES + l];
            pos += 4;
            x1 = 2.0 * to_double(a1, b1) - 1.0;
            x2 = 2.0 * to_double(a2, b2) - 1.0;
            r2 = x1 * x1 + x2 * x2;
          } while (r2 >= 1.0 || r2 == 0.0);
          const double f = sqrt(-2.0 * log(r2) / r2);
          out[(first + l) * (size_t)k + j] = f * x2; j++;
          if (j < k) { out[(first + l) * (size_t)k + j] = f * x1; j++; }
        }
      }
    }
  }
  free(b);
}

int g_threads = 0;

int allowed_threads() {
  if (g_threads > 0) return g_threads;
  /* the affinity mask, bounded by the cgroup's CPU quota (cgroup v2 cpu.max, v1 cfs quota): more runnable threads than
   * the quota allows only take turns */
  cpu_set_t set;
  int n = 1;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
  double quota = 0.0;
  if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[32]; double per = 0.0;
    if (fscanf(f, "%31s %lf", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0.0) quota = atof(q) / per;
    fclose(f);
  } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
    double q = 0.0, per = 0.0;
    if (fscanf(g, "%lf", &q) == 1 && q > 0.0) {
      if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%lf", &per) == 1 && per > 0.0) quota = q / per; fclose(h); }
    }
    fclose(g);
  }
  if (quota >= 1.0) n = std::min(n, (int)(quota + 0.5));
  return std::max(1, std::min(n, 16));
}

int run(Kind kind, const int64_t *seeds, size_t n, int k, double *out) {
  if (!seeds || !out || k <= 0) return n == 0 ? 0 : -1;
  for (size_t i = 0; i < n; i++)
    if (seeds[i] < 0 || (kind != PY_RANDOM && seeds[i] > 0xffffffffLL)) return -1;
  const size_t blocks = (n + LANES - 1) / LANES;
  genrand_19650218();   /* built before any thread reads it */
  const size_t threads = std::min<size_t>((size_t)allowed_threads(), std::max<size_t>(1, blocks / 64));
  if (threads <= 1) { run_blocks(kind, seeds, n, k, out, 0, blocks); return 0; }
  std::vector<std::thread> pool;
  for (size_t t = 0; t < threads; t++) {
    const size_t b0 = blocks * t / threads, b1 = blocks * (t + 1) / threads;
    pool.emplace_back(run_blocks, kind, seeds, n, k, out, b0, b1);
  }
  for (auto &th : pool) th.join();
  return 0;
}

}  // namespace

extern "C" {
int npb_seed_py_random(const int64_t *seeds, size_t n, int k, double *out) { return run(PY_RANDOM, seeds, n, k, out); }
int npb_seed_np_random(const int64_t *seeds, size_t n, int k, double *out) { return run(NP_RANDOM, seeds, n, k, out); }
int npb_seed_np_gauss(const int64_t *seeds, size_t n, int k, double *out) { return run(NP_GAUSS, seeds, n, k, out); }
int npb_seed_set_threads(int threads) { g_threads = threads > 0 ? std::min(threads, 64) : 0; return allowed_threads(); }
}
