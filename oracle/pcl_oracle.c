/*
 * pcl_oracle.c -- CPU restatement of the PCL ICP hot path.  TEST INFRASTRUCTURE ONLY
 * (see pcl_oracle.h for the contract, parity status and who may load this).
 *
 * Build: gcc -O2 -std=c11 -fopenmp -ffp-contract=off -fno-fast-math -shared -fPIC (oracle/Makefile)
 * All citations are file:line relative to /root/reference.
 */
#define _GNU_SOURCE
#include "pcl_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static int finite3(const float* p) { return isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]); }

/* FLANN L2_Simple: result = 0; for each dim: diff = a-b; result += diff*diff (float). */
static inline float l2_simple(const float* a, const float* b) {
  float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  float r = dx * dx;
  r = r + dy * dy;
  r = r + dz * dz;
  return r;
}

/* ------------------------------------------------------------------------------------------ */
/* k-NN result set: ascending by (d2, idx); bounded size k.                                    */
typedef struct {
  int k, n;
  float* d;
  int32_t* i;
} knn_set;

static inline int key_less(float d1, int32_t i1, float d2, int32_t i2) {
  return d1 < d2 || (d1 == d2 && i1 < i2);
}

static inline void knn_insert(knn_set* s, float d, int32_t idx) {
  if (s->n == s->k && !key_less(d, idx, s->d[s->k - 1], s->i[s->k - 1])) return;
  int pos = (s->n < s->k) ? s->n : s->k - 1;
  while (pos > 0 && key_less(d, idx, s->d[pos - 1], s->i[pos - 1])) {
    s->d[pos] = s->d[pos - 1];
    s->i[pos] = s->i[pos - 1];
    --pos;
  }
  s->d[pos] = d;
  s->i[pos] = idx;
  if (s->n < s->k) s->n++;
}

int orc_knn_bruteforce(const float* tgt, int64_t nt, int ts, const float* qry, int64_t nq, int qs,
                       int k, int32_t* out_idx, float* out_d2, int nthreads) {
  int64_t nvalid = 0;
  for (int64_t j = 0; j < nt; ++j) nvalid += finite3(tgt + j * ts);
  int keff = (k < nvalid) ? k : (int)nvalid; /* kdtree_flann.hpp:241-242 */
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 64)
  for (int64_t i = 0; i < nq; ++i) {
    knn_set s = {keff, 0, out_d2 + i * k, out_idx + i * k};
    const float* q = qry + i * qs;
    if (finite3(q) && keff > 0) {
      for (int64_t j = 0; j < nt; ++j) {
        const float* p = tgt + j * ts;
        if (!finite3(p)) continue;
        knn_insert(&s, l2_simple(q, p), (int32_t)j);
      }
    }
    for (int c = s.n; c < k; ++c) {
      out_idx[i * k + c] = -1;
      out_d2[i * k + c] = INFINITY;
    }
  }
  return keff;
}

/* ------------------------------------------------------------------------------------------ */
/* Exact kd-tree.                                                                               */
typedef struct {
  float bmin[3], bmax[3]; /* tight bbox of the points below this node */
  int32_t left, right;    /* children (internal) or [begin,end) into pts (leaf) */
  int32_t is_leaf;
} kd_node;

struct orc_kdtree {
  int64_t n;        /* finite points */
  float* pts;       /* reordered xyz, 3 floats per point */
  int32_t* orig;    /* original index of reordered point */
  kd_node* nodes;
  int64_t nnodes, cap;
};

#define KD_LEAF 15

static int32_t kd_new_node(orc_kdtree* t) {
  if (t->nnodes == t->cap) {
    t->cap = t->cap ? t->cap * 2 : 1024;
    t->nodes = (kd_node*)realloc(t->nodes, (size_t)t->cap * sizeof(kd_node));
  }
  return (int32_t)t->nnodes++;
}

static void kd_bbox(const orc_kdtree* t, int64_t b, int64_t e, float* mn, float* mx) {
  for (int d = 0; d < 3; ++d) {
    mn[d] = FLT_MAX;
    mx[d] = -FLT_MAX;
  }
  for (int64_t i = b; i < e; ++i)
    for (int d = 0; d < 3; ++d) {
      float v = t->pts[3 * i + d];
      if (v < mn[d]) mn[d] = v;
      if (v > mx[d]) mx[d] = v;
    }
}

static void kd_swap(orc_kdtree* t, int64_t a, int64_t b) {
  float tmp[3];
  memcpy(tmp, t->pts + 3 * a, sizeof tmp);
  memcpy(t->pts + 3 * a, t->pts + 3 * b, sizeof tmp);
  memcpy(t->pts + 3 * b, tmp, sizeof tmp);
  int32_t o = t->orig[a];
  t->orig[a] = t->orig[b];
  t->orig[b] = o;
}

static int32_t kd_build(orc_kdtree* t, int64_t b, int64_t e) {
  int32_t id = kd_new_node(t);
  float mn[3], mx[3];
  kd_bbox(t, b, e, mn, mx);
  memcpy(t->nodes[id].bmin, mn, sizeof mn);
  memcpy(t->nodes[id].bmax, mx, sizeof mx);
  if (e - b <= KD_LEAF) {
    t->nodes[id].is_leaf = 1;
    t->nodes[id].left = (int32_t)b;
    t->nodes[id].right = (int32_t)e;
    return id;
  }
  int dim = 0;
  float span = mx[0] - mn[0];
  for (int d = 1; d < 3; ++d)
    if (mx[d] - mn[d] > span) {
      span = mx[d] - mn[d];
      dim = d;
    }
  int64_t mid;
  if (span <= 0.0f) {
    mid = (b + e) / 2; /* all points identical */
  } else {
    float split = 0.5f * (mn[dim] + mx[dim]);
    int64_t lo = b, hi = e - 1;
    while (lo <= hi) {
      while (lo <= hi && t->pts[3 * lo + dim] < split) ++lo;
      while (lo <= hi && t->pts[3 * hi + dim] >= split) --hi;
      if (lo < hi) {
        kd_swap(t, lo, hi);
        ++lo;
        --hi;
      }
    }
    mid = lo;
    if (mid == b || mid == e) mid = (b + e) / 2; /* cannot happen with span > 0, safety */
  }
  int32_t l = kd_build(t, b, mid);
  int32_t r = kd_build(t, mid, e);
  t->nodes[id].is_leaf = 0;
  t->nodes[id].left = l;
  t->nodes[id].right = r;
  return id;
}

orc_kdtree* orc_kdtree_build(const float* pts, int64_t n, int stride) {
  orc_kdtree* t = (orc_kdtree*)calloc(1, sizeof *t);
  t->pts = (float*)malloc((size_t)(n > 0 ? n : 1) * 3 * sizeof(float));
  t->orig = (int32_t*)malloc((size_t)(n > 0 ? n : 1) * sizeof(int32_t));
  int64_t m = 0;
  for (int64_t i = 0; i < n; ++i) {
    const float* p = pts + i * stride;
    if (!finite3(p)) continue; /* kdtree_flann.hpp:443-452 */
    t->pts[3 * m] = p[0];
    t->pts[3 * m + 1] = p[1];
    t->pts[3 * m + 2] = p[2];
    t->orig[m] = (int32_t)i;
    ++m;
  }
  t->n = m;
  if (m > 0) kd_build(t, 0, m);
  return t;
}

void orc_kdtree_free(orc_kdtree* t) {
  if (!t) return;
  free(t->pts);
  free(t->orig);
  free(t->nodes);
  free(t);
}

int64_t orc_kdtree_size(const orc_kdtree* t) { return t->n; }

/* Monotone lower bound: for every point c inside the box, l2_simple(q,c) >= box_lb(q,box) as
 * floats (rounding is monotone, same operation order). */
static inline float box_lb(const float* q, const float* mn, const float* mx) {
  float g[3];
  for (int d = 0; d < 3; ++d) {
    float a = mn[d] - q[d], b = q[d] - mx[d];
    float m = a > b ? a : b;
    g[d] = m > 0.0f ? m : 0.0f;
  }
  float r = g[0] * g[0];
  r = r + g[1] * g[1];
  r = r + g[2] * g[2];
  return r;
}

static void kd_search(const orc_kdtree* t, int32_t id, const float* q, knn_set* s) {
  const kd_node* nd = &t->nodes[id];
  if (nd->is_leaf) {
    for (int32_t i = nd->left; i < nd->right; ++i)
      knn_insert(s, l2_simple(q, t->pts + 3 * (int64_t)i), t->orig[i]);
    return;
  }
  const kd_node* L = &t->nodes[nd->left];
  const kd_node* R = &t->nodes[nd->right];
  float dl = box_lb(q, L->bmin, L->bmax), dr = box_lb(q, R->bmin, R->bmax);
  int32_t first = nd->left, second = nd->right;
  float d1 = dl, d2 = dr;
  if (dr < dl) {
    first = nd->right;
    second = nd->left;
    d1 = dr;
    d2 = dl;
  }
  if (s->n < s->k || d1 <= s->d[s->k - 1]) kd_search(t, first, q, s);
  if (s->n < s->k || d2 <= s->d[s->k - 1]) kd_search(t, second, q, s);
}

int orc_kdtree_knn(const orc_kdtree* t, const float* qry, int64_t nq, int qs, int k,
                   int32_t* out_idx, float* out_d2, int nthreads) {
  int keff = (k < t->n) ? k : (int)t->n;
  if (nthreads < 1) nthreads = 1;
  /* The reference's loop over the source points (impl/correspondence_estimation.hpp:163-191) is a plain `omp parallel
   * for`: static, contiguous shares.  Contiguous runs matter when the queries are spatially ordered (a thread then stays
   * inside one part of the tree); chunks of ~1/16 of a thread's share keep that and still balance uneven queries. */
  int64_t chunk = nq / ((int64_t)nthreads * 16);
  if (chunk < 64) chunk = 64;
  if (chunk > 16384) chunk = 16384;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, chunk)
  for (int64_t i = 0; i < nq; ++i) {
    knn_set s = {keff, 0, out_d2 + i * k, out_idx + i * k};
    const float* q = qry + i * qs;
    if (finite3(q) && keff > 0) kd_search(t, 0, q, &s);
    for (int c = s.n; c < k; ++c) {
      out_idx[i * k + c] = -1;
      out_d2[i * k + c] = INFINITY;
    }
  }
  return keff;
}

/* ------------------------------------------------------------------------------------------ */
/* nn / dd: scratch of ns entries each, or NULL (allocated here).  A caller that runs many iterations brings its own: fresh
 * pages first touched inside the parallel loop are page faults under every thread at once (the whole process shares one
 * address-space lock), which at 256 threads costs more than the search. */
static int64_t correspondences_with(const orc_kdtree* t, const float* src, int64_t ns, int ss, double max_dist,
                                    int32_t* out_q, int32_t* out_m, float* out_d2, int nthreads, int32_t* nn,
                                    float* dd) {
  /* correspondence_estimation.hpp:161 */
  const double max_dist_sqr = max_dist * max_dist;
  int32_t* own_nn = NULL;
  float* own_dd = NULL;
  if (nn == NULL) nn = own_nn = (int32_t*)malloc((size_t)(ns > 0 ? ns : 1) * sizeof(int32_t));
  if (dd == NULL) dd = own_dd = (float*)malloc((size_t)(ns > 0 ? ns : 1) * sizeof(float));
  orc_kdtree_knn(t, src, ns, ss, 1, nn, dd, nthreads);
  int64_t c = 0;
  for (int64_t i = 0; i < ns; ++i) {
    if (nn[i] < 0) continue;                    /* non-finite source point (:173-174) */
    if ((double)dd[i] > max_dist_sqr) continue; /* :176 */
    out_q[c] = (int32_t)i;
    out_m[c] = nn[i];
    out_d2[c] = dd[i];
    ++c;
  }
  free(own_nn);
  free(own_dd);
  return c;
}

int64_t orc_correspondences(const orc_kdtree* t, const float* src, int64_t ns, int ss,
                            double max_dist, int32_t* out_q, int32_t* out_m, float* out_d2,
                            int nthreads) {
  return correspondences_with(t, src, ns, ss, max_dist, out_q, out_m, out_d2, nthreads, NULL, NULL);
}

/* Registration::getFitnessScore, registration/include/pcl/registration/impl/registration.hpp:132-168.
 * input_transformed = transformPointCloud(input, T) (Transformer::se3 order); for every finite point
 * the 1-NN squared distance is added (double, ascending point order) when it is <= max_range
 * (compared as given: squared distance against max_range, :157); returns sum/nr or DBL_MAX. */
double orc_fitness_score(const orc_kdtree* t, const float* src, int64_t ns, int ss, const float* T,
                         double max_range, int64_t* out_nr, int nthreads) {
  float* cur = (float*)calloc((size_t)(ns > 0 ? ns : 1) * 4, sizeof(float));
  for (int64_t i = 0; i < ns; ++i) {
    cur[4 * i] = src[i * ss];
    cur[4 * i + 1] = src[i * ss + 1];
    cur[4 * i + 2] = src[i * ss + 2];
    cur[4 * i + 3] = 1.0f;
  }
  orc_transform_cloud(T, 1, cur, 4, cur, 4, NULL, 0, NULL, 0, ns);
  int32_t* nn = (int32_t*)malloc((size_t)(ns > 0 ? ns : 1) * sizeof(int32_t));
  float* dd = (float*)malloc((size_t)(ns > 0 ? ns : 1) * sizeof(float));
  orc_kdtree_knn(t, cur, ns, 4, 1, nn, dd, nthreads);
  double fitness = 0.0;
  int64_t nr = 0;
  for (int64_t i = 0; i < ns; ++i) {
    if (nn[i] < 0) continue; /* non-finite point (:151-152) */
    if ((double)dd[i] <= max_range) {
      fitness += (double)dd[i];
      ++nr;
    }
  }
  free(cur);
  free(nn);
  free(dd);
  if (out_nr) *out_nr = nr;
  return nr > 0 ? fitness / (double)nr : DBL_MAX;
}

int64_t orc_reciprocal_correspondences(const orc_kdtree* tgt_tree, const orc_kdtree* src_tree,
                                       const float* src, int64_t ns, int ss, const float* tgt,
                                       int ts, double max_dist, int32_t* out_q, int32_t* out_m,
                                       float* out_d2, int nthreads) {
  const double max_dist_sqr = max_dist * max_dist;
  int32_t* nn = (int32_t*)malloc((size_t)(ns > 0 ? ns : 1) * sizeof(int32_t));
  float* dd = (float*)malloc((size_t)(ns > 0 ? ns : 1) * sizeof(float));
  orc_kdtree_knn(tgt_tree, src, ns, ss, 1, nn, dd, nthreads);
  int64_t c = 0;
  for (int64_t i = 0; i < ns; ++i) {
    if (nn[i] < 0 || (double)dd[i] > max_dist_sqr) continue;
    int32_t ri;
    float rd;
    orc_kdtree_knn(src_tree, tgt + (int64_t)nn[i] * ts, 1, ts, 1, &ri, &rd, 1);
    if ((double)rd > max_dist_sqr || ri != (int32_t)i) continue; /* :265-266 */
    out_q[c] = (int32_t)i;
    out_m[c] = nn[i];
    out_d2[c] = dd[i];
    ++c;
  }
  free(nn);
  free(dd);
  return c;
}

/* ------------------------------------------------------------------------------------------ */
/* 6x6 inverse by Gauss-Jordan with partial pivoting (Eigen Matrix<double,6,6>::inverse() is a
 * PartialPivLU solve), then x = inv * b.                                                        */
static void solve6_inverse(const double* A, const double* b, double* x) {
  double M[6][12];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) {
      M[i][j] = A[i * 6 + j];
      M[i][6 + j] = (i == j) ? 1.0 : 0.0;
    }
  for (int c = 0; c < 6; ++c) {
    int piv = c;
    for (int r = c + 1; r < 6; ++r)
      if (fabs(M[r][c]) > fabs(M[piv][c])) piv = r;
    if (piv != c)
      for (int j = 0; j < 12; ++j) {
        double tmp = M[c][j];
        M[c][j] = M[piv][j];
        M[piv][j] = tmp;
      }
    double p = M[c][c];
    for (int j = 0; j < 12; ++j) M[c][j] /= p;
    for (int r = 0; r < 6; ++r) {
      if (r == c) continue;
      double f = M[r][c];
      if (f == 0.0) continue;
      for (int j = 0; j < 12; ++j) M[r][j] -= f * M[c][j];
    }
  }
  for (int i = 0; i < 6; ++i) {
    double s = 0.0;
    for (int j = 0; j < 6; ++j) s += M[i][6 + j] * b[j];
    x[i] = s;
  }
}

/* constructTransformationMatrix, transformation_estimation_point_to_plane_lls.hpp:132-163 */
static void construct_T(double alpha, double beta, double gamma, double tx, double ty, double tz,
                        float* T) {
  memset(T, 0, 16 * sizeof(float));
  T[0] = (float)(cos(gamma) * cos(beta));
  T[1] = (float)(-sin(gamma) * cos(alpha) + cos(gamma) * sin(beta) * sin(alpha));
  T[2] = (float)(sin(gamma) * sin(alpha) + cos(gamma) * sin(beta) * cos(alpha));
  T[4] = (float)(sin(gamma) * cos(beta));
  T[5] = (float)(cos(gamma) * cos(alpha) + sin(gamma) * sin(beta) * sin(alpha));
  T[6] = (float)(-cos(gamma) * sin(alpha) + sin(gamma) * sin(beta) * cos(alpha));
  T[8] = (float)(-sin(beta));
  T[9] = (float)(cos(beta) * sin(alpha));
  T[10] = (float)(cos(beta) * cos(alpha));
  T[3] = (float)tx;
  T[7] = (float)ty;
  T[11] = (float)tz;
  T[15] = 1.0f;
}

void orc_lls_solve(const double* s, float* T) {
  /* s[0..20]: upper triangle in the order of :213-233, s[21..26]: ATb */
  double A[36], b[6], x[6];
  int k = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 6; ++j) {
      A[i * 6 + j] = s[k];
      A[j * 6 + i] = s[k]; /* mirror, :247-261 */
      ++k;
    }
  for (int i = 0; i < 6; ++i) b[i] = s[21 + i];
  solve6_inverse(A, b, x); /* :264 */
  construct_T(x[0], x[1], x[2], x[3], x[4], x[5], T);
}

int64_t orc_lls_point_to_plane(const float* src, int ss, const float* tgt, int ts,
                               const float* nrm, int ns_, const int32_t* q, const int32_t* m,
                               int64_t npairs, double* sums27, float* T) {
  double s[27];
  memset(s, 0, sizeof s);
  int64_t used = 0;
  for (int64_t p = 0; p < npairs; ++p) {
    const float* S = src + (int64_t)(q ? q[p] : p) * ss;
    const float* D = tgt + (int64_t)(m ? m[p] : p) * ts;
    const float* N = nrm + (int64_t)(m ? m[p] : p) * ns_;
    if (!finite3(S) || !finite3(D) || !finite3(N)) continue; /* :182-190 */
    const float sx = S[0], sy = S[1], sz = S[2];
    const float dx = D[0], dy = D[1], dz = D[2];
    const float nx = N[0], ny = N[1], nz = N[2];
    /* float arithmetic, stored as double (:202-204) */
    double a = (double)(nz * sy - ny * sz);
    double b = (double)(nx * sz - nz * sx);
    double c = (double)(ny * sx - nx * sy);
    s[0] += a * a;
    s[1] += a * b;
    s[2] += a * c;
    s[3] += a * nx;
    s[4] += a * ny;
    s[5] += a * nz;
    s[6] += b * b;
    s[7] += b * c;
    s[8] += b * nx;
    s[9] += b * ny;
    s[10] += b * nz;
    s[11] += c * c;
    s[12] += c * nx;
    s[13] += c * ny;
    s[14] += c * nz;
    s[15] += (double)(nx * nx); /* float products (:228-233) */
    s[16] += (double)(nx * ny);
    s[17] += (double)(nx * nz);
    s[18] += (double)(ny * ny);
    s[19] += (double)(ny * nz);
    s[20] += (double)(nz * nz);
    /* :235 -- float expression evaluated left to right */
    float df = nx * dx + ny * dy + nz * dz - nx * sx - ny * sy - nz * sz;
    double d = (double)df;
    s[21] += a * d;
    s[22] += b * d;
    s[23] += c * d;
    s[24] += nx * d;
    s[25] += ny * d;
    s[26] += nz * d;
    ++used;
  }
  if (sums27) memcpy(sums27, s, sizeof s);
  if (T) orc_lls_solve(s, T);
  return used;
}

/* ------------------------------------------------------------------------------------------ */
/* 3x3 SVD by one-sided Jacobi (double), singular values sorted descending.                     */
static void svd3(const double A[9], double U[9], double S[3], double V[9]) {
  double B[9]; /* columns become U*S */
  memcpy(B, A, sizeof B);
  for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int r = 0; r < 3; ++r) {
          alpha += B[r * 3 + p] * B[r * 3 + p];
          beta += B[r * 3 + q] * B[r * 3 + q];
          gamma += B[r * 3 + p] * B[r * 3 + q];
        }
        if (gamma == 0.0) continue;
        off += fabs(gamma) / sqrt(alpha * beta + 1e-300);
        double zeta = (beta - alpha) / (2.0 * gamma);
        double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int r = 0; r < 3; ++r) {
          double bp = B[r * 3 + p], bq = B[r * 3 + q];
          B[r * 3 + p] = c * bp - s * bq;
          B[r * 3 + q] = s * bp + c * bq;
          double vp = V[r * 3 + p], vq = V[r * 3 + q];
          V[r * 3 + p] = c * vp - s * vq;
          V[r * 3 + q] = s * vp + c * vq;
        }
      }
    if (off < 1e-15) break;
  }
  for (int c = 0; c < 3; ++c) S[c] = sqrt(B[c] * B[c] + B[3 + c] * B[3 + c] + B[6 + c] * B[6 + c]);
  /* sort descending */
  int ord[3] = {0, 1, 2};
  for (int i = 0; i < 2; ++i)
    for (int j = i + 1; j < 3; ++j)
      if (S[ord[j]] > S[ord[i]]) {
        int tmp = ord[i];
        ord[i] = ord[j];
        ord[j] = tmp;
      }
  double Bs[9], Vs[9], Ss[3];
  for (int c = 0; c < 3; ++c) {
    Ss[c] = S[ord[c]];
    for (int r = 0; r < 3; ++r) {
      Bs[r * 3 + c] = B[r * 3 + ord[c]];
      Vs[r * 3 + c] = V[r * 3 + ord[c]];
    }
  }
  memcpy(S, Ss, sizeof Ss);
  memcpy(V, Vs, sizeof Vs);
  /* U columns: normalised B columns; complete degenerate columns by cross products */
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) U[r * 3 + c] = (S[c] > 1e-300) ? Bs[r * 3 + c] / S[c] : 0.0;
  double tol = 1e-12 * (S[0] > 0 ? S[0] : 1.0);
  if (S[0] <= tol) {
    for (int i = 0; i < 9; ++i) U[i] = (i % 4 == 0) ? 1.0 : 0.0;
  } else {
    if (S[1] <= tol) { /* pick any unit vector orthogonal to u0 */
      double u0[3] = {U[0], U[3], U[6]};
      int mi = 0;
      if (fabs(u0[1]) < fabs(u0[mi])) mi = 1;
      if (fabs(u0[2]) < fabs(u0[mi])) mi = 2;
      double e[3] = {0, 0, 0};
      e[mi] = 1.0;
      double dot = u0[mi];
      double v[3] = {e[0] - dot * u0[0], e[1] - dot * u0[1], e[2] - dot * u0[2]};
      double nv = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
      U[1] = v[0] / nv;
      U[4] = v[1] / nv;
      U[7] = v[2] / nv;
    }
    if (S[2] <= tol) { /* u2 = u0 x u1 */
      U[2] = U[3] * U[7] - U[6] * U[4];
      U[5] = U[6] * U[1] - U[0] * U[7];
      U[8] = U[0] * U[4] - U[3] * U[1];
    }
  }
}

static double det3(const double* M) {
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
         M[2] * (M[3] * M[7] - M[4] * M[6]);
}

/* eigen.hpp:716-736: Rt = U * diag(S) * V^T; t = dst_mean - R * src_mean */
static void umeyama_finish(const double sigma[9], const double src_mean[3], const double dst_mean[3],
                           float* T) {
  double U[9], S[3], V[9];
  svd3(sigma, U, S, V);
  double sgn[3] = {1, 1, 1};
  if (det3(U) * det3(V) < 0) sgn[2] = -1; /* eigen.hpp:719-720 */
  double R[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += U[i * 3 + k] * sgn[k] * V[j * 3 + k];
      R[i * 3 + j] = s;
    }
  memset(T, 0, 16 * sizeof(float));
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) T[i * 4 + j] = (float)R[i * 3 + j];
    double rs = R[i * 3] * src_mean[0] + R[i * 3 + 1] * src_mean[1] + R[i * 3 + 2] * src_mean[2];
    T[i * 4 + 3] = (float)(dst_mean[i] - rs);
  }
  T[15] = 1.0f;
}

int64_t orc_umeyama(const float* src, int ss, const float* tgt, int ts, const int32_t* q,
                    const int32_t* m, int64_t npairs, int acc_double, float* T) {
  /* eigen.hpp:696-712: means, demean, sigma = 1/n * dst_demean * src_demean^T */
  double sm[3] = {0, 0, 0}, dm[3] = {0, 0, 0}, sigma[9];
  const int64_t n = npairs;
  if (n == 0) return 0;
  if (acc_double) {
    for (int64_t p = 0; p < n; ++p) {
      const float* S = src + (int64_t)(q ? q[p] : p) * ss;
      const float* D = tgt + (int64_t)(m ? m[p] : p) * ts;
      for (int d = 0; d < 3; ++d) {
        sm[d] += S[d];
        dm[d] += D[d];
      }
    }
    for (int d = 0; d < 3; ++d) {
      sm[d] /= (double)n;
      dm[d] /= (double)n;
    }
    memset(sigma, 0, sizeof sigma);
    for (int64_t p = 0; p < n; ++p) {
      const float* S = src + (int64_t)(q ? q[p] : p) * ss;
      const float* D = tgt + (int64_t)(m ? m[p] : p) * ts;
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) sigma[i * 3 + j] += (D[i] - dm[i]) * (S[j] - sm[j]);
    }
    for (int i = 0; i < 9; ++i) sigma[i] /= (double)n;
  } else {
    float fsm[3] = {0, 0, 0}, fdm[3] = {0, 0, 0}, fs[9] = {0};
    const float one_over_n = 1.0f / (float)n;
    for (int64_t p = 0; p < n; ++p) {
      const float* S = src + (int64_t)(q ? q[p] : p) * ss;
      const float* D = tgt + (int64_t)(m ? m[p] : p) * ts;
      for (int d = 0; d < 3; ++d) {
        fsm[d] = fsm[d] + S[d];
        fdm[d] = fdm[d] + D[d];
      }
    }
    for (int d = 0; d < 3; ++d) {
      fsm[d] = fsm[d] * one_over_n;
      fdm[d] = fdm[d] * one_over_n;
    }
    for (int64_t p = 0; p < n; ++p) {
      const float* S = src + (int64_t)(q ? q[p] : p) * ss;
      const float* D = tgt + (int64_t)(m ? m[p] : p) * ts;
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) fs[i * 3 + j] = fs[i * 3 + j] + (D[i] - fdm[i]) * (S[j] - fsm[j]);
    }
    for (int i = 0; i < 9; ++i) sigma[i] = (double)(one_over_n * fs[i]);
    for (int d = 0; d < 3; ++d) {
      sm[d] = fsm[d];
      dm[d] = fdm[d];
    }
  }
  umeyama_finish(sigma, sm, dm, T);
  return n;
}

void orc_umeyama_from_sums(const double* s, double count, float* T) {
  double sm[3], dm[3], sigma[9];
  for (int d = 0; d < 3; ++d) {
    sm[d] = s[d] / count;
    dm[d] = s[3 + d] / count;
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) sigma[i * 3 + j] = s[6 + i * 3 + j] / count - dm[i] * sm[j];
  umeyama_finish(sigma, sm, dm, T);
}

/* ------------------------------------------------------------------------------------------ */
void orc_transform_cloud(const float* T, int order, const float* in, int is_, float* out, int os,
                         const float* nrm_in, int nis, float* nrm_out, int nos, int64_t n) {
  for (int64_t i = 0; i < n; ++i) {
    const float* p = in + i * is_;
    float* o = out + i * os;
    if (!finite3(p)) continue; /* icp.hpp:97-98 / transforms.hpp:381-384 */
    const float x = p[0], y = p[1], z = p[2];
    float r[3];
    for (int k = 0; k < 3; ++k) {
      const float* row = T + 4 * k;
      if (order == 0)
        r[k] = ((row[0] * x + row[1] * y) + row[2] * z) + row[3] * 1.0f;
      else
        r[k] = row[0] * x + (row[1] * y + (row[2] * z + row[3]));
    }
    o[0] = r[0];
    o[1] = r[1];
    o[2] = r[2];
    if (nrm_in && nrm_out) {
      const float* nn = nrm_in + i * nis;
      float* no = nrm_out + i * nos;
      if (order == 0 && !finite3(nn)) continue; /* icp.hpp:84-85 */
      const float a = nn[0], b = nn[1], c = nn[2];
      float s[3];
      for (int k = 0; k < 3; ++k) {
        const float* row = T + 4 * k;
        if (order == 0)
          s[k] = (row[0] * a + row[1] * b) + row[2] * c;
        else
          s[k] = row[0] * a + (row[1] * b + row[2] * c);
      }
      no[0] = s[0];
      no[1] = s[1];
      no[2] = s[2];
    }
  }
}

void orc_mat4_mul(const float* A, const float* B, float* C) {
  float R[16];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      R[i * 4 + j] = ((A[i * 4 + 0] * B[0 * 4 + j] + A[i * 4 + 1] * B[1 * 4 + j]) +
                      A[i * 4 + 2] * B[2 * 4 + j]) +
                     A[i * 4 + 3] * B[3 * 4 + j];
  memcpy(C, R, sizeof R);
}

/* ------------------------------------------------------------------------------------------ */
void orc_convergence_init(orc_convergence* c) {
  c->max_iterations = 1000;
  c->failure_after_max_iter = 0;
  c->rotation_threshold = 0.99999;
  c->translation_threshold = 3e-4 * 3e-4;
  c->mse_threshold_relative = 0.00001;
  c->mse_threshold_absolute = 1e-12;
  c->max_iterations_similar_transforms = 0;
  c->iterations_similar_transforms = 0;
  c->correspondences_prev_mse = DBL_MAX;
  c->correspondences_cur_mse = DBL_MAX;
  c->convergence_state = ORC_NOT_CONVERGED;
}

int orc_convergence_has_converged(orc_convergence* c, int iterations, const float* T, double mse) {
  if (c->convergence_state != ORC_NOT_CONVERGED) { /* :52-56 */
    c->iterations_similar_transforms = 0;
    c->convergence_state = ORC_NOT_CONVERGED;
  }
  int is_similar = 0;
  if (iterations >= c->max_iterations) { /* :65-71 */
    if (!c->failure_after_max_iter) {
      c->convergence_state = ORC_ITERATIONS;
      return 1;
    }
    c->convergence_state = ORC_FAILURE_AFTER_MAX_ITERATIONS;
  }
  /* :75-79 -- Scalar=float coefficients, summed in float then promoted by the 0.5* double */
  double cos_angle = 0.5 * (double)(T[0] + T[5] + T[10] - 1);
  double translation_sqr = (double)(T[3] * T[3] + T[7] * T[7] + T[11] * T[11]);
  if (cos_angle >= c->rotation_threshold && translation_sqr <= c->translation_threshold) {
    if (c->iterations_similar_transforms >= c->max_iterations_similar_transforms) {
      c->convergence_state = ORC_TRANSFORM;
      return 1;
    }
    is_similar = 1;
  }
  c->correspondences_cur_mse = mse;
  if (fabs(c->correspondences_cur_mse - c->correspondences_prev_mse) < c->mse_threshold_absolute) {
    if (c->iterations_similar_transforms >= c->max_iterations_similar_transforms) {
      c->convergence_state = ORC_ABS_MSE;
      return 1;
    }
    is_similar = 1;
  }
  if (fabs(c->correspondences_cur_mse - c->correspondences_prev_mse) / c->correspondences_prev_mse <
      c->mse_threshold_relative) {
    if (c->iterations_similar_transforms >= c->max_iterations_similar_transforms) {
      c->convergence_state = ORC_REL_MSE;
      return 1;
    }
    is_similar = 1;
  }
  if (is_similar)
    ++c->iterations_similar_transforms;
  else
    c->iterations_similar_transforms = 0;
  c->correspondences_prev_mse = c->correspondences_cur_mse;
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
void orc_icp_params_default(orc_icp_params* p) {
  p->max_iterations = 10;
  p->max_correspondence_distance = sqrt(DBL_MAX);
  p->transformation_epsilon = 0.0;
  p->transformation_rotation_epsilon = 0.0;
  p->euclidean_fitness_epsilon = -DBL_MAX;
  p->min_number_correspondences = 3;
  p->mode = 0;
  p->acc_double = 0;
  p->nthreads = 1;
  p->use_reciprocal = 0;
}

int orc_icp_align(const orc_kdtree* tgt_tree, const float* tgt, int ts, const float* tgt_nrm,
                  int tns, const float* src, int64_t ns, int ss, const float* guess,
                  const orc_icp_params* p, orc_convergence* conv, orc_icp_result* r,
                  float* per_iter_T, int32_t* per_iter_match) {
  const double t0 = now_s();
  static const float I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  const int order = p->mode == 1 ? 1 : 0;
  memset(r, 0, sizeof *r);
  float* cur = (float*)malloc((size_t)(ns > 0 ? ns : 1) * 4 * sizeof(float)); /* input_transformed */
  for (int64_t i = 0; i < ns; ++i) {
    cur[4 * i] = src[i * ss];
    cur[4 * i + 1] = src[i * ss + 1];
    cur[4 * i + 2] = src[i * ss + 2];
    cur[4 * i + 3] = 1.0f;
  }
  float final_T[16], Tk[16];
  memcpy(final_T, guess ? guess : I4, sizeof final_T); /* icp.hpp:123 */
  if (guess && memcmp(guess, I4, sizeof I4) != 0)       /* :126-130 */
    orc_transform_cloud(guess, order, cur, 4, cur, 4, NULL, 0, NULL, 0, ns);
  memcpy(Tk, I4, sizeof Tk);
  int32_t* cq = (int32_t*)malloc((size_t)(ns > 0 ? ns : 1) * sizeof(int32_t));
  int32_t* cm = (int32_t*)malloc((size_t)(ns > 0 ? ns : 1) * sizeof(int32_t));
  float* cd = (float*)malloc((size_t)(ns > 0 ? ns : 1) * sizeof(float));
  /* scratch of the search, touched once here (see correspondences_with) */
  int32_t* s_nn = (int32_t*)calloc((size_t)(ns > 0 ? ns : 1), sizeof(int32_t));
  float* s_dd = (float*)calloc((size_t)(ns > 0 ? ns : 1), sizeof(float));
  if (s_nn) memset(s_nn, 0xFF, (size_t)(ns > 0 ? ns : 1) * sizeof(int32_t));
  if (s_dd) memset(s_dd, 0, (size_t)(ns > 0 ? ns : 1) * sizeof(float));
  orc_kdtree* src_tree = NULL;
  /* :157-161 */
  conv->max_iterations = p->max_iterations;
  conv->mse_threshold_relative = p->euclidean_fitness_epsilon;
  conv->translation_threshold = p->transformation_epsilon;
  if (p->transformation_rotation_epsilon > 0)
    conv->rotation_threshold = p->transformation_rotation_epsilon;
  int nr_iterations = 0, converged = 0;
  do {
    const double ts0 = now_s();
    int64_t nc;
    if (p->use_reciprocal) {
      src_tree = orc_kdtree_build(cur, ns, 4);
      nc = orc_reciprocal_correspondences(tgt_tree, src_tree, cur, ns, 4, tgt, ts,
                                          p->max_correspondence_distance, cq, cm, cd, p->nthreads);
      orc_kdtree_free(src_tree);
    } else {
      nc = correspondences_with(tgt_tree, cur, ns, 4, p->max_correspondence_distance, cq, cm, cd, p->nthreads, s_nn,
                                s_dd);
    }
    r->seconds_search += now_s() - ts0;
    if (per_iter_match) {
      int32_t* row = per_iter_match + (int64_t)nr_iterations * ns;
      for (int64_t i = 0; i < ns; ++i) row[i] = -1;
      for (int64_t c = 0; c < nc; ++c) row[cq[c]] = cm[c];
    }
    r->last_num_correspondences = nc;
    if (nc < p->min_number_correspondences) { /* :204-213 */
      conv->convergence_state = ORC_NO_CORRESPONDENCES;
      converged = 0;
      break;
    }
    if (p->mode == 1)
      orc_lls_point_to_plane(cur, 4, tgt, ts, tgt_nrm, tns, cq, cm, nc, NULL, Tk);
    else
      orc_umeyama(cur, 4, tgt, ts, cq, cm, nc, p->acc_double, Tk);
    if (per_iter_T) memcpy(per_iter_T + 16 * (int64_t)nr_iterations, Tk, sizeof Tk);
    orc_transform_cloud(Tk, order, cur, 4, cur, 4, NULL, 0, NULL, 0, ns); /* :220 */
    orc_mat4_mul(Tk, final_T, final_T);                                   /* :223 */
    ++nr_iterations;
    double mse = 0; /* calculateMSE, default_convergence_criteria.h:262-270 */
    for (int64_t c = 0; c < nc; ++c) mse += cd[c];
    mse /= (double)nc;
    r->last_mse = mse;
    converged = orc_convergence_has_converged(conv, nr_iterations, Tk, mse);
  } while (conv->convergence_state == ORC_NOT_CONVERGED);
  memcpy(r->final_transformation, final_T, sizeof final_T);
  r->nr_iterations = nr_iterations;
  r->converged = converged;
  r->convergence_state = conv->convergence_state;
  r->seconds_total = now_s() - t0;
  free(cur);
  free(cq);
  free(cm);
  free(cd);
  free(s_nn);
  free(s_dd);
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* TransformationEstimationSymmetricPointToPlaneLLS,
 * registration/include/pcl/registration/impl/transformation_estimation_symmetric_point_to_plane_lls.hpp:149-197.
 * Per pair (float, Scalar = float): n = n1 + n2, or n1 - n2 when enforce_same_direction and n1.n2 < 0
 * (:167-177); skipped unless p, q, n are finite (:179-182); v = [(p+q) x n ; n]; ATA += v v^T
 * (rankUpdate, :186); ATb += v * ((q-p).n) (:188).  acc_double = 0: sums in float, sequential (the
 * reference sums in float inside Eigen, order unspecified); acc_double = 1: float terms, double sums.
 * sums27: upper triangle row-major [21] + ATb [6]. */
void orc_symmetric_solve(const double* s, float* T);
int64_t orc_lls_symmetric(const float* src, int ss, const float* src_nrm, int sns, const float* tgt,
                          int ts, const float* tgt_nrm, int tns, const int32_t* q, const int32_t* m,
                          int64_t npairs, int enforce_same_direction, int acc_double, double* sums27,
                          float* T) {
  double sd[27];
  float sf[27];
  memset(sd, 0, sizeof sd);
  memset(sf, 0, sizeof sf);
  int64_t used = 0;
  for (int64_t k = 0; k < npairs; ++k) {
    const int64_t iq = q ? q[k] : k, im = m ? m[k] : k;
    const float* P = src + iq * ss;
    const float* N1 = src_nrm + iq * sns;
    const float* Q = tgt + im * ts;
    const float* N2 = tgt_nrm + im * tns;
    float n[3];
    const float dot12 = (N1[0] * N2[0] + N1[1] * N2[1]) + N1[2] * N2[2];
    if (enforce_same_direction && !(dot12 >= 0.0f)) {
      n[0] = N1[0] - N2[0];
      n[1] = N1[1] - N2[1];
      n[2] = N1[2] - N2[2];
    } else {
      n[0] = N1[0] + N2[0];
      n[1] = N1[1] + N2[1];
      n[2] = N1[2] + N2[2];
    }
    if (!finite3(P) || !finite3(Q) || !finite3(n)) continue;
    const float sx = P[0] + Q[0], sy = P[1] + Q[1], sz = P[2] + Q[2];
    float v[6];
    v[0] = sy * n[2] - sz * n[1];
    v[1] = sz * n[0] - sx * n[2];
    v[2] = sx * n[1] - sy * n[0];
    v[3] = n[0];
    v[4] = n[1];
    v[5] = n[2];
    const float dx = Q[0] - P[0], dy = Q[1] - P[1], dz = Q[2] - P[2];
    const float r = (dx * n[0] + dy * n[1]) + dz * n[2];
    int c = 0;
    for (int i = 0; i < 6; ++i)
      for (int j = i; j < 6; ++j) {
        const float t = v[i] * v[j];
        sd[c] += (double)t;
        sf[c] += t;
        ++c;
      }
    for (int i = 0; i < 6; ++i) {
      const float t = v[i] * r;
      sd[21 + i] += (double)t;
      sf[21 + i] += t;
    }
    ++used;
  }
  if (!acc_double)
    for (int i = 0; i < 27; ++i) sd[i] = (double)sf[i];
  if (sums27) memcpy(sums27, sd, sizeof sd);
  if (T) orc_symmetric_solve(sd, T);
  return used;
}

/* x = ATA^-1 ATb (:193, LDLT in the reference; Gaussian elimination with partial pivoting in double
 * here), then constructTransformationMatrix (:128-147):
 * T = Rz Ry Rx * Translation(t) * Rz Ry Rx = [R R | R t], R = Rz(x2) Ry(x1) Rx(x0). */
void orc_symmetric_solve(const double* s, float* T) {
  double A[6][7];
  int c = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 6; ++j) {
      A[i][j] = s[c];
      A[j][i] = s[c];
      ++c;
    }
  for (int i = 0; i < 6; ++i) A[i][6] = s[21 + i];
  double x[6];
  int singular = 0;
  for (int col = 0; col < 6; ++col) {
    int piv = col;
    for (int r = col + 1; r < 6; ++r)
      if (fabs(A[r][col]) > fabs(A[piv][col])) piv = r;
    if (!(fabs(A[piv][col]) > 0.0)) {
      singular = 1;
      break;
    }
    if (piv != col)
      for (int k = 0; k < 7; ++k) {
        const double t = A[col][k];
        A[col][k] = A[piv][k];
        A[piv][k] = t;
      }
    for (int r = col + 1; r < 6; ++r) {
      const double f = A[r][col] / A[col][col];
      for (int k = col; k < 7; ++k) A[r][k] -= f * A[col][k];
    }
  }
  for (int i = 5; i >= 0 && !singular; --i) {
    double v = A[i][6];
    for (int k = i + 1; k < 6; ++k) v -= A[i][k] * x[k];
    x[i] = v / A[i][i];
  }
  if (singular)
    for (int i = 0; i < 6; ++i) x[i] = NAN;
  const double ca = cos(x[0]), sa = sin(x[0]), cb = cos(x[1]), sb = sin(x[1]), cg = cos(x[2]), sg = sin(x[2]);
  const double R[3][3] = {{cg * cb, -sg * ca + cg * sb * sa, sg * sa + cg * sb * ca},
                          {sg * cb, cg * ca + sg * sb * sa, -cg * sa + sg * sb * ca},
                          {-sb, cb * sa, cb * ca}};
  memset(T, 0, 16 * sizeof(float));
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) {
      double v = 0.0;
      for (int k = 0; k < 3; ++k) v += R[i][k] * R[k][j];
      T[4 * i + j] = (float)v;
    }
    T[4 * i + 3] = (float)(R[i][0] * x[3] + R[i][1] * x[4] + R[i][2] * x[5]);
  }
  T[15] = 1.0f;
}

/* ------------------------------------------------------------------------------------------ */
/* GeneralizedIterativeClosestPoint::computeCovariances, registration/include/pcl/registration/impl/gicp.hpp
 * :70-147.  Per point: k nearest neighbours (:101), float differences to the query accumulated in
 * double (:104-120), mean and covariance (:122-129), JacobiSVD of the symmetric 3x3 (:132; restated as a
 * cyclic Jacobi eigen-iteration: singular values = |eigenvalues|, U = eigenvectors), singular values
 * replaced by (1, 1, epsilon) (:134-143) => cov = I - (1 - eps) u3 u3^T.  out: 9 doubles per point.
 * Parity note: the reference's tests hold no golden vector for these matrices; this function is pinned
 * by its properties (tests/test_oracle_golden.py: eigenvalues {1, 1, eps}, u3 = known plane normal). */
static void orc_jacobi3(double A[3][3], double V[3][3]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 64; ++sweep) {
    const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
    if (off < 1e-300) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (A[p][q] == 0.0) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) {
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq;
          A[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) {
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk;
          A[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; ++k) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
  }
}

int orc_gicp_covariances(const orc_kdtree* t, const float* cloud, int64_t n, int cs, int k, double eps,
                         double* out, int nthreads) {
  if (k < 1 || (int64_t)k > orc_kdtree_size(t)) return -1; /* :77-83 */
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
  {
    int32_t* idx = (int32_t*)malloc((size_t)k * sizeof(int32_t));
    float* d2 = (float*)malloc((size_t)k * sizeof(float));
#pragma omp for schedule(dynamic, 256)
    for (int64_t i = 0; i < n; ++i) {
      const float* p = cloud + i * cs;
      double* o = out + 9 * i;
      if (!finite3(p)) {
        for (int c = 0; c < 9; ++c) o[c] = NAN;
        continue;
      }
      orc_kdtree_knn(t, p, 1, cs, k, idx, d2, 1);
      double mean[3] = {0, 0, 0}, cov[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
      for (int j = 0; j < k; ++j) {
        const float* q = cloud + (int64_t)idx[j] * cs;
        const double ptx = (double)(q[0] - p[0]), pty = (double)(q[1] - p[1]), ptz = (double)(q[2] - p[2]);
        mean[0] += ptx;
        mean[1] += pty;
        mean[2] += ptz;
        cov[0][0] += ptx * ptx;
        cov[1][0] += pty * ptx;
        cov[1][1] += pty * pty;
        cov[2][0] += ptz * ptx;
        cov[2][1] += ptz * pty;
        cov[2][2] += ptz * ptz;
      }
      for (int a = 0; a < 3; ++a) mean[a] /= (double)k;
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b <= a; ++b) {
          cov[a][b] /= (double)k;
          cov[a][b] -= mean[a] * mean[b];
          cov[b][a] = cov[a][b];
        }
      double V[3][3];
      orc_jacobi3(cov, V);
      const double w0 = fabs(cov[0][0]), w1 = fabs(cov[1][1]), w2 = fabs(cov[2][2]);
      const int m = (w0 <= w1 && w0 <= w2) ? 0 : ((w1 <= w2) ? 1 : 2);
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) o[3 * a + b] = (a == b ? 1.0 : 0.0) - (1.0 - eps) * V[a][m] * V[b][m];
    }
    free(idx);
    free(d2);
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
unsigned orc_mean_and_covariance(const float* cloud, int cs, const int32_t* indices, int n,
                                 float* cov, float* centroid) {
  /* centroid.hpp:587-648 (the !is_dense branch; identical to the dense one on finite data) */
  float accu[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  float K[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const float* p = cloud + (int64_t)indices[i] * cs;
    if (finite3(p)) {
      K[0] = p[0];
      K[1] = p[1];
      K[2] = p[2];
      break;
    }
  }
  unsigned count = 0;
  for (int i = 0; i < n; ++i) {
    const float* p = cloud + (int64_t)indices[i] * cs;
    if (!finite3(p)) continue;
    ++count;
    float x = p[0] - K[0], y = p[1] - K[1], z = p[2] - K[2];
    accu[0] += x * x;
    accu[1] += x * y;
    accu[2] += x * z;
    accu[3] += y * y;
    accu[4] += y * z;
    accu[5] += z * z;
    accu[6] += x;
    accu[7] += y;
    accu[8] += z;
  }
  if (count != 0) {
    const float fc = (float)count;
    for (int i = 0; i < 9; ++i) accu[i] = accu[i] / fc;
    centroid[0] = accu[6] + K[0];
    centroid[1] = accu[7] + K[1];
    centroid[2] = accu[8] + K[2];
    centroid[3] = 1;
    cov[0] = accu[0] - accu[6] * accu[6];
    cov[1] = accu[1] - accu[6] * accu[7];
    cov[2] = accu[2] - accu[6] * accu[8];
    cov[4] = accu[3] - accu[7] * accu[7];
    cov[5] = accu[4] - accu[7] * accu[8];
    cov[8] = accu[5] - accu[8] * accu[8];
    cov[3] = cov[1];
    cov[6] = cov[2];
    cov[7] = cov[5];
  }
  return count;
}

/* eigen.hpp:52-65 */
static void compute_roots2(float b, float c, float* roots) {
  roots[0] = 0.0f;
  float d = (float)((double)(b * b) - 4.0 * (double)c);
  if (d < 0.0f) d = 0.0f;
  float sd = sqrtf(d);
  roots[2] = 0.5f * (b + sd);
  roots[1] = 0.5f * (b - sd);
}

/* eigen.hpp:68-128; m is row-major symmetric 3x3 */
static void compute_roots(const float* m, float* roots) {
#define M(i, j) m[(i)*3 + (j)]
  float c0 = M(0, 0) * M(1, 1) * M(2, 2) + 2.0f * M(0, 1) * M(0, 2) * M(1, 2) -
             M(0, 0) * M(1, 2) * M(1, 2) - M(1, 1) * M(0, 2) * M(0, 2) - M(2, 2) * M(0, 1) * M(0, 1);
  float c1 = M(0, 0) * M(1, 1) - M(0, 1) * M(0, 1) + M(0, 0) * M(2, 2) - M(0, 2) * M(0, 2) +
             M(1, 1) * M(2, 2) - M(1, 2) * M(1, 2);
  float c2 = M(0, 0) + M(1, 1) + M(2, 2);
#undef M
  if (fabsf(c0) < FLT_EPSILON) {
    compute_roots2(c2, c1, roots);
  } else {
    const float s_inv3 = (float)(1.0 / 3.0);
    const float s_sqrt3 = sqrtf(3.0f);
    float c2_over_3 = c2 * s_inv3;
    float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
    if (a_over_3 > 0.0f) a_over_3 = 0.0f;
    float half_b = 0.5f * (c0 + c2_over_3 * (2.0f * c2_over_3 * c2_over_3 - c1));
    float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
    if (q > 0.0f) q = 0.0f;
    float rho = sqrtf(-a_over_3);
    float theta = atan2f(sqrtf(-q), half_b) * s_inv3;
    float cos_theta = cosf(theta);
    float sin_theta = sinf(theta);
    roots[0] = c2_over_3 + 2.0f * rho * cos_theta;
    roots[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
    roots[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
    float t;
    if (roots[0] >= roots[1]) {
      t = roots[0];
      roots[0] = roots[1];
      roots[1] = t;
    }
    if (roots[1] >= roots[2]) {
      t = roots[1];
      roots[1] = roots[2];
      roots[2] = t;
      if (roots[0] >= roots[1]) {
        t = roots[0];
        roots[0] = roots[1];
        roots[1] = t;
      }
    }
    if (roots[0] <= 0) compute_roots2(c2, c1, roots);
  }
}

static void cross3(const float* a, const float* b, float* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

/* eigen.hpp:272-290 */
static void largest_eigvec(const float* sm, float* v) {
  float cp[3][3];
  cross3(sm + 0, sm + 3, cp[0]);
  cross3(sm + 0, sm + 6, cp[1]);
  cross3(sm + 3, sm + 6, cp[2]);
  int best = 0;
  float len = -1.0f;
  for (int i = 0; i < 3; ++i) {
    float l = sqrtf((cp[i][0] * cp[i][0] + cp[i][1] * cp[i][1]) + cp[i][2] * cp[i][2]);
    if (l > len) {
      len = l;
      best = i;
    }
  }
  for (int d = 0; d < 3; ++d) v[d] = cp[best][d] / len;
}

/* Eigen unitOrthogonal for 3-vectors (used only in the degenerate branch eigen.hpp:316-318) */
static void unit_orthogonal(const float* s, float* o) {
  const float prec = 1e-5f; /* NumTraits<float>::dummy_precision() */
  int x_small = fabsf(s[0]) <= prec * fabsf(s[2]);
  int y_small = fabsf(s[1]) <= prec * fabsf(s[2]);
  if (!x_small || !y_small) {
    float inv = 1.0f / sqrtf(s[0] * s[0] + s[1] * s[1]);
    o[0] = -s[1] * inv;
    o[1] = s[0] * inv;
    o[2] = 0.0f;
  } else {
    float inv = 1.0f / sqrtf(s[1] * s[1] + s[2] * s[2]);
    o[0] = 0.0f;
    o[1] = -s[2] * inv;
    o[2] = s[1] * inv;
  }
}

/* eigen.hpp:295-325 */
static void eigen33(const float* mat, float* eigenvalue, float* vec) {
  float scale = 0.0f;
  for (int i = 0; i < 9; ++i)
    if (fabsf(mat[i]) > scale) scale = fabsf(mat[i]);
  if (scale <= FLT_MIN) scale = 1.0f;
  float sm[9];
  for (int i = 0; i < 9; ++i) sm[i] = mat[i] / scale;
  float ev[3];
  compute_roots(sm, ev);
  *eigenvalue = ev[0] * scale;
  if ((ev[1] - ev[0]) > FLT_EPSILON) {
    sm[0] -= ev[0];
    sm[4] -= ev[0];
    sm[8] -= ev[0];
    largest_eigvec(sm, vec);
  } else if ((ev[2] - ev[0]) > FLT_EPSILON) {
    sm[0] -= ev[2];
    sm[4] -= ev[2];
    sm[8] -= ev[2];
    float tmp[3];
    largest_eigvec(sm, tmp);
    unit_orthogonal(tmp, vec);
  } else {
    vec[0] = 1.0f;
    vec[1] = 0.0f;
    vec[2] = 0.0f;
  }
}

void orc_solve_plane_parameters(const float* cov, float* nx, float* ny, float* nz,
                                float* curvature) {
  float ev, v[3];
  eigen33(cov, &ev, v);
  *nx = v[0];
  *ny = v[1];
  *nz = v[2];
  float eig_sum = cov[0] + cov[4] + cov[8]; /* feature.hpp:84 */
  if (eig_sum != 0)
    *curvature = fabsf(ev / eig_sum);
  else
    *curvature = 0;
}

/* computeFeature loops over indices_ (features/include/pcl/features/impl/normal_3d.hpp:59,74): with
 * `indices` == NULL every point of the cloud is a query, else only cloud[indices[j]], output row j. */
int64_t orc_normals_knn_indices(const orc_kdtree* t, const float* cloud, int64_t n, int cs, int k,
                                const float* vp, const int32_t* indices, int64_t n_indices, float* out,
                                int32_t* out_knn, int nthreads) {
  int64_t nan_count = 0;
  if (nthreads < 1) nthreads = 1;
  const int64_t nq = indices ? n_indices : n;
#pragma omp parallel num_threads(nthreads) reduction(+ : nan_count)
  {
    int32_t* idx = (int32_t*)malloc((size_t)k * sizeof(int32_t));
    float* d2 = (float*)malloc((size_t)k * sizeof(float));
#pragma omp for schedule(dynamic, 256)
    for (int64_t j = 0; j < nq; ++j) {
      const int64_t i = indices ? (int64_t)indices[j] : j;
      const float* p = cloud + i * cs;
      float* o = out + 4 * j;
      int found = 0;
      if (finite3(p)) found = orc_kdtree_knn(t, p, 1, cs, k, idx, d2, 1);
      if (out_knn)
        for (int c = 0; c < k; ++c) out_knn[j * k + c] = (finite3(p) && c < found) ? idx[c] : -1;
      float cov[9], cen[4];
      /* normal_3d.hpp:79-87, normal_3d.h:308-322 */
      if (!finite3(p) || found == 0 || found < 3 ||
          orc_mean_and_covariance(cloud, cs, idx, found, cov, cen) == 0) {
        o[0] = o[1] = o[2] = o[3] = NAN;
        ++nan_count;
        continue;
      }
      orc_solve_plane_parameters(cov, &o[0], &o[1], &o[2], &o[3]);
      /* flipNormalTowardsViewpoint, normal_3d.h:169-188 */
      float vx = vp[0] - p[0], vy = vp[1] - p[1], vz = vp[2] - p[2];
      float cos_theta = (vx * o[0] + vy * o[1] + vz * o[2]);
      if (cos_theta < 0) {
        o[0] *= -1;
        o[1] *= -1;
        o[2] *= -1;
      }
    }
    free(idx);
    free(d2);
  }
  return nan_count;
}

int64_t orc_normals_knn(const orc_kdtree* t, const float* cloud, int64_t n, int cs, int k,
                        const float* vp, float* out, int32_t* out_knn, int nthreads) {
  return orc_normals_knn_indices(t, cloud, n, cs, k, vp, NULL, 0, out, out_knn, nthreads);
}

/* Feature::setSearchSurface (features/include/pcl/features/impl/feature.hpp:104-118,195-229): the tree holds the
 * SURFACE cloud, the queries come from another cloud (`input_`, restricted to `indices` when given).  computeFeature
 * (impl/normal_3d.hpp:48-95) then searches the surface around every query, fits the plane to the surface points
 * found (normal_3d.h:308-322: computePointNormal(*surface_, nn_indices, ...)) and flips the normal towards the
 * viewpoint as seen from the QUERY point (normal_3d.hpp:66,87: input_->points[idx]). */
int64_t orc_normals_knn_queries(const orc_kdtree* t, const float* surface, int ss, const float* queries, int64_t nq_cloud,
                                int qs, const int32_t* indices, int64_t n_indices, int k, const float* vp, float* out,
                                int32_t* out_knn, int nthreads) {
  int64_t nan_count = 0;
  if (nthreads < 1) nthreads = 1;
  const int64_t nq = indices ? n_indices : nq_cloud;
#pragma omp parallel num_threads(nthreads) reduction(+ : nan_count)
  {
    int32_t* idx = (int32_t*)malloc((size_t)k * sizeof(int32_t));
    float* d2 = (float*)malloc((size_t)k * sizeof(float));
#pragma omp for schedule(dynamic, 256)
    for (int64_t j = 0; j < nq; ++j) {
      const int64_t i = indices ? (int64_t)indices[j] : j;
      const float* p = queries + i * qs;
      float* o = out + 4 * j;
      int found = 0;
      if (finite3(p)) found = orc_kdtree_knn(t, p, 1, qs, k, idx, d2, 1);
      if (out_knn)
        for (int c = 0; c < k; ++c) out_knn[j * k + c] = (finite3(p) && c < found) ? idx[c] : -1;
      float cov[9], cen[4];
      if (!finite3(p) || found < 3 || orc_mean_and_covariance(surface, ss, idx, found, cov, cen) == 0) {
        o[0] = o[1] = o[2] = o[3] = NAN;
        ++nan_count;
        continue;
      }
      orc_solve_plane_parameters(cov, &o[0], &o[1], &o[2], &o[3]);
      float vx = vp[0] - p[0], vy = vp[1] - p[1], vz = vp[2] - p[2]; /* normal_3d.h:169-188 */
      float cos_theta = (vx * o[0] + vy * o[1] + vz * o[2]);
      if (cos_theta < 0) {
        o[0] *= -1;
        o[1] *= -1;
        o[2] *= -1;
      }
    }
    free(idx);
    free(d2);
  }
  return nan_count;
}

/* ------------------------------------------------------------------------------------------ */
typedef struct {
  uint32_t idx;
  int32_t pt;
} vox_pair;

static int vox_limit_field(int limits) {
  const int f = (limits >> 8) & 0xFF;
  return f ? f - 1 : 2;
}

static int vox_cmp(const void* a, const void* b) {
  const vox_pair* x = (const vox_pair*)a;
  const vox_pair* y = (const vox_pair*)b;
  if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1;
  return x->pt < y->pt ? -1 : (x->pt > y->pt ? 1 : 0); /* stable: ascending input index */
}

int64_t orc_voxelgrid(const float* cloud, int64_t n, int cs, const float* leaf,
                      unsigned min_points_per_voxel, int has_limits, double lim_min, double lim_max,
                      float* out, int32_t* out_voxel_ids) {
  float inv[3]; /* voxel_grid.h:279-282 */
  for (int d = 0; d < 3; ++d) inv[d] = 1.0f / leaf[d];
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  const float fmin_ = (float)lim_min, fmax_ = (float)lim_max; /* voxel_grid.hpp:615 casts */
  for (int64_t i = 0; i < n; ++i) {
    const float* p = cloud + i * cs;
    if (has_limits) { /* voxel_grid.hpp:535-553: the field (bits 8..15: 1 + float position, 0 = z), negative = bit 1 */
      const float v = p[vox_limit_field(has_limits)];
      if (has_limits & 2) {
        if ((v < fmax_) && (v > fmin_)) continue;
      } else {
        if ((v > fmax_) || (v < fmin_)) continue;
      }
    }
    if (!finite3(p)) continue;
    for (int d = 0; d < 3; ++d) {
      if (p[d] < mn[d]) mn[d] = p[d];
      if (p[d] > mx[d]) mx[d] = p[d];
    }
  }
  /* :620-629 */
  int64_t dx = (int64_t)((mx[0] - mn[0]) * inv[0]) + 1;
  int64_t dy = (int64_t)((mx[1] - mn[1]) * inv[1]) + 1;
  int64_t dz = (int64_t)((mx[2] - mn[2]) * inv[2]) + 1;
  if (dx * dy * dz > (int64_t)INT32_MAX) return -1;
  int min_b[3], max_b[3], div_b[3];
  for (int d = 0; d < 3; ++d) { /* :632-640 */
    min_b[d] = (int)floorf(mn[d] * inv[d]);
    max_b[d] = (int)floorf(mx[d] * inv[d]);
    div_b[d] = max_b[d] - min_b[d] + 1;
  }
  const int mul[3] = {1, div_b[0], div_b[0] * div_b[1]};
  vox_pair* v = (vox_pair*)malloc((size_t)(n > 0 ? n : 1) * sizeof(vox_pair));
  int64_t m = 0;
  for (int64_t i = 0; i < n; ++i) {
    const float* p = cloud + i * cs;
    if (!finite3(p)) continue;
    if (has_limits) { /* :675-695, double limits against the float value */
      const double v = (double)p[vox_limit_field(has_limits)];
      if (has_limits & 2) {
        if ((v < lim_max) && (v > lim_min)) continue;
      } else {
        if ((v > lim_max) || (v < lim_min)) continue;
      }
    }
    /* :713-718 */
    int i0 = (int)(floorf(p[0] * inv[0]) - (float)min_b[0]);
    int i1 = (int)(floorf(p[1] * inv[1]) - (float)min_b[1]);
    int i2 = (int)(floorf(p[2] * inv[2]) - (float)min_b[2]);
    int idx = i0 * mul[0] + i1 * mul[1] + i2 * mul[2];
    v[m].idx = (uint32_t)idx;
    v[m].pt = (int32_t)i;
    ++m;
  }
  qsort(v, (size_t)m, sizeof(vox_pair), vox_cmp);
  int64_t total = 0, index = 0;
  while (index < m) { /* :735-813 */
    int64_t i = index + 1;
    while (i < m && v[i].idx == v[index].idx) ++i;
    if ((uint64_t)(i - index) >= min_points_per_voxel) {
      float sx = 0, sy = 0, sz = 0; /* accumulators.hpp:76-81 */
      for (int64_t li = index; li < i; ++li) {
        const float* p = cloud + (int64_t)v[li].pt * cs;
        sx += p[0];
        sy += p[1];
        sz += p[2];
      }
      const float cnt = (float)(i - index);
      out[4 * total + 0] = sx / cnt;
      out[4 * total + 1] = sy / cnt;
      out[4 * total + 2] = sz / cnt;
      out[4 * total + 3] = 1.0f;
      if (out_voxel_ids) out_voxel_ids[total] = (int32_t)v[index].idx;
      ++total;
    }
    index = i;
  }
  free(v);
  return total;
}
