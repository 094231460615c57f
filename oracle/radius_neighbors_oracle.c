/* TEST INFRASTRUCTURE ONLY -- the product path never imports, links or calls this.
 *
 * CPU restatement (plain C) of the reference's fixed-radius neighbour search
 *   geotransformer/extensions/cpu/radius_neighbors/radius_neighbors_cpu.cpp:3-91
 * whose per-pair metric / acceptance / ordering rules live in the vendored
 * nanoflann 1.3.0:
 *   metric   ((dx*dx + dy*dy) + dz*dz, fp32, accumulated from 0, no FMA)
 *            geotransformer/extensions/extra/nanoflann/nanoflann.hpp:423-446
 *   accept   strict  d < r*r                          nanoflann.hpp:249-253
 *   order    ascending d (std::sort, IndexDist_Sorter) nanoflann.hpp:208-214, :1287
 * The reference walks a kd-tree; the result set depends only on the three
 * rules above, so this restatement enumerates candidates
 * with a uniform grid (cell edge >= radius) or by brute force and applies the
 * same rules.  Equal-distance ties are ordered by ascending support index here
 * (the reference's tie order depends on kd-tree traversal and is not defined).
 *
 * Pinned against: tests/golden/ext_*.npz (outputs of the reference itself, made
 * by tests/golden/gen_golden_ext.py through oracle/_ref) -- see
 * tests/test_oracle_ext.py.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (oracle/Makefile).  Do NOT add
 * -march=native / -ffast-math: the arithmetic must stay separate fp32 mul/add.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  float d;
  int64_t i;
} hit_t;

static int hit_cmp(const void* a, const void* b) {
  const hit_t* x = (const hit_t*)a;
  const hit_t* y = (const hit_t*)b;
  if (x->d < y->d) return -1;
  if (x->d > y->d) return 1;
  return (x->i > y->i) - (x->i < y->i);
}

/* nanoflann.hpp:432-440 -- result += diff*diff, dims in order x,y,z, from 0.f */
static inline float sqdist(const float* a, const float* b) {
  float r = 0.0f;
  float d0 = a[0] - b[0];
  r += d0 * d0;
  float d1 = a[1] - b[1];
  r += d1 * d1;
  float d2 = a[2] - b[2];
  r += d2 * d2;
  return r;
}

typedef struct {
  hit_t* v;
  int64_t n, cap;
} hitvec_t;

static void hv_push(hitvec_t* h, float d, int64_t i) {
  if (h->n == h->cap) {
    h->cap = h->cap ? h->cap * 2 : 64;
    h->v = (hit_t*)realloc(h->v, sizeof(hit_t) * (size_t)h->cap);
  }
  h->v[h->n].d = d;
  h->v[h->n].i = i;
  h->n++;
}

/* Per-query neighbour lists in CSR form (row_ptr has nq+1 entries). */
typedef struct {
  int64_t* row_ptr;
  int64_t* idx; /* support index LOCAL to its batch element */
  int64_t total;
} csr_t;

static void search_batch_brute(const float* q, int64_t nq, const float* s, int64_t ns, float r2,
                               hitvec_t* all, int64_t* row_n) {
  hitvec_t tmp = {0, 0, 0};
  for (int64_t i = 0; i < nq; ++i) {
    tmp.n = 0;
    for (int64_t j = 0; j < ns; ++j) {
      float d = sqdist(q + 3 * i, s + 3 * j);
      if (d < r2) hv_push(&tmp, d, j);
    }
    qsort(tmp.v, (size_t)tmp.n, sizeof(hit_t), hit_cmp);
    for (int64_t k = 0; k < tmp.n; ++k) hv_push(all, tmp.v[k].d, tmp.v[k].i);
    row_n[i] = tmp.n;
  }
  free(tmp.v);
}

static void search_batch_grid(const float* q, int64_t nq, const float* s, int64_t ns, float radius,
                              float r2, hitvec_t* all, int64_t* row_n) {
  if (ns == 0) {
    for (int64_t i = 0; i < nq; ++i) row_n[i] = 0;
    return;
  }
  /* bounding box of the supports */
  double mn[3], mx[3];
  for (int k = 0; k < 3; ++k) mn[k] = mx[k] = s[k];
  for (int64_t j = 0; j < ns; ++j)
    for (int k = 0; k < 3; ++k) {
      double v = s[3 * j + k];
      if (v < mn[k]) mn[k] = v;
      if (v > mx[k]) mx[k] = v;
    }
  /* cell edge: >= radius with margin; coarsen so the dense grid stays small */
  double cell = (double)radius * (1.0 + 1.0 / 1024.0);
  if (!(cell > 0.0)) cell = 1.0;
  const double max_cells = (double)(ns < 1024 ? 4096 : 4 * ns);
  int64_t dim[3];
  for (;;) {
    double tot = 1.0;
    for (int k = 0; k < 3; ++k) {
      double e = floor((mx[k] - mn[k]) / cell) + 1.0;
      if (e < 1.0) e = 1.0;
      dim[k] = (int64_t)(e < 4.0e6 ? e : 4.0e6);
      tot *= e;
    }
    if (tot <= max_cells) break;
    cell *= 1.26; /* ~ x2 volume */
  }
  const int64_t ncell = dim[0] * dim[1] * dim[2];
  int64_t* start = (int64_t*)calloc((size_t)ncell + 1, sizeof(int64_t));
  int64_t* cell_of = (int64_t*)malloc(sizeof(int64_t) * (size_t)ns);
  int64_t* order = (int64_t*)malloc(sizeof(int64_t) * (size_t)ns);
  for (int64_t j = 0; j < ns; ++j) {
    int64_t c[3];
    for (int k = 0; k < 3; ++k) {
      int64_t ci = (int64_t)floor(((double)s[3 * j + k] - mn[k]) / cell);
      if (ci < 0) ci = 0;
      if (ci >= dim[k]) ci = dim[k] - 1;
      c[k] = ci;
    }
    cell_of[j] = c[0] + dim[0] * (c[1] + dim[1] * c[2]);
    start[cell_of[j] + 1]++;
  }
  for (int64_t c = 0; c < ncell; ++c) start[c + 1] += start[c];
  int64_t* cur = (int64_t*)malloc(sizeof(int64_t) * (size_t)ncell);
  memcpy(cur, start, sizeof(int64_t) * (size_t)ncell);
  for (int64_t j = 0; j < ns; ++j) order[cur[cell_of[j]]++] = j;

  hitvec_t tmp = {0, 0, 0};
  for (int64_t i = 0; i < nq; ++i) {
    tmp.n = 0;
    int64_t lo[3], hi[3];
    int empty = 0;
    for (int k = 0; k < 3; ++k) {
      double u = floor(((double)q[3 * i + k] - mn[k]) / cell);
      double l = u - 1.0, h = u + 1.0;
      if (h < 0.0 || l > (double)(dim[k] - 1)) empty = 1;
      if (l < 0.0) l = 0.0;
      if (h > (double)(dim[k] - 1)) h = (double)(dim[k] - 1);
      lo[k] = (int64_t)l;
      hi[k] = (int64_t)h;
    }
    if (!empty) {
      for (int64_t cz = lo[2]; cz <= hi[2]; ++cz)
        for (int64_t cy = lo[1]; cy <= hi[1]; ++cy) {
          int64_t base = dim[0] * (cy + dim[1] * cz);
          for (int64_t p = start[base + lo[0]]; p < start[base + hi[0] + 1]; ++p) {
            int64_t j = order[p];
            float d = sqdist(q + 3 * i, s + 3 * j);
            if (d < r2) hv_push(&tmp, d, j);
          }
        }
    }
    qsort(tmp.v, (size_t)tmp.n, sizeof(hit_t), hit_cmp);
    for (int64_t k = 0; k < tmp.n; ++k) hv_push(all, tmp.v[k].d, tmp.v[k].i);
    row_n[i] = tmp.n;
  }
  free(tmp.v);
  free(start);
  free(cell_of);
  free(order);
  free(cur);
}

/* radius_neighbors_cpu.cpp:3-91 restated.  mode 0 = grid candidates, 1 = brute force.
 * Returns max_count; *out is malloc'ed (nq * max_count int64, pad = ns total,
 * radius_neighbors_cpu.cpp:83-85); release with oracle_free().  Empty query batch
 * elements are tolerated here (the reference is undefined for them). */
int64_t oracle_radius_neighbors(const float* q, int64_t nq, const float* s, int64_t ns,
                                const int64_t* q_lengths, const int64_t* s_lengths, int64_t batch,
                                float radius, int mode, int64_t** out) {
  const float r2 = radius * radius; /* radius_neighbors_cpu.cpp:12 (fp32) */
  hitvec_t all = {0, 0, 0};
  int64_t* row_n = (int64_t*)calloc((size_t)(nq > 0 ? nq : 1), sizeof(int64_t));
  int64_t* row_s0 = (int64_t*)calloc((size_t)(nq > 0 ? nq : 1), sizeof(int64_t));
  int64_t q0 = 0, s0 = 0;
  for (int64_t b = 0; b < batch; ++b) {
    if (mode == 1)
      search_batch_brute(q + 3 * q0, q_lengths[b], s + 3 * s0, s_lengths[b], r2, &all, row_n + q0);
    else
      search_batch_grid(q + 3 * q0, q_lengths[b], s + 3 * s0, s_lengths[b], radius, r2, &all,
                        row_n + q0);
    for (int64_t i = 0; i < q_lengths[b]; ++i) row_s0[q0 + i] = s0;
    q0 += q_lengths[b];
    s0 += s_lengths[b];
  }
  int64_t max_count = 0; /* radius_neighbors_cpu.cpp:59-61 */
  for (int64_t i = 0; i < nq; ++i)
    if (row_n[i] > max_count) max_count = row_n[i];
  int64_t* o = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nq * max_count > 0 ? nq * max_count : 1));
  int64_t p = 0;
  for (int64_t i = 0; i < nq; ++i) {
    for (int64_t j = 0; j < max_count; ++j)
      o[i * max_count + j] = j < row_n[i] ? all.v[p + j].i + row_s0[i] : ns;
    p += row_n[i];
  }
  free(all.v);
  free(row_n);
  free(row_s0);
  *out = o;
  return max_count;
}

void oracle_free(void* p) { free(p); }
