/* icp_pcd.c -- the path end to end through the C ABI only (plain C99, no C++/Python):
 *   load two PCD files -> index the target -> k-NN normals -> point-to-plane ICP -> fitness score ->
 *   write the registered source as binary_compressed PCD.
 * Build:  gcc -std=c99 -O2 -Iinclude examples/icp_pcd.c -Lpcl_amd -lpclhip -Wl,-rpath,$PWD/pcl_amd -lm -o icp_pcd
 * Usage:  ./icp_pcd source.pcd target.pcd [out.pcd] [max_correspondence_distance] [max_iterations]
 * This is what pcl's doc/tutorials/content/sources/iterative_closest_point/iterative_closest_point.cpp does with
 * pcl::io::loadPCDFile + pcl::IterativeClosestPointWithNormals. */
#include <float.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pclhip.h"

#define CHECK(call, ctx)                                                              \
  do {                                                                                \
    pclhip_status st__ = (call);                                                      \
    if (st__ != PCLHIP_OK) {                                                          \
      fprintf(stderr, "%s failed (%d): %s\n", #call, (int)st__, pclhip_last_error(ctx)); \
      return 1;                                                                       \
    }                                                                                 \
  } while (0)

static float* load_cloud(const char* path, uint64_t* n) {
  pclhip_pcd_info info;
  if (pclhip_pcd_read_header(path, &info) != PCLHIP_OK) {
    fprintf(stderr, "%s: %s\n", path, pclhip_last_error(NULL));
    return NULL;
  }
  float* pts = (float*)malloc((size_t)(info.points ? info.points : 1) * 16); /* PointXYZ records: x y z 1 */
  int dense = 1;
  if (!pts || pclhip_pcd_read(path, pts, 16, 0, info.points, n, &dense) != PCLHIP_OK) {
    fprintf(stderr, "%s: %s\n", path, pclhip_last_error(NULL));
    free(pts);
    return NULL;
  }
  printf("%s: %llu points (%s, %s)\n", path, (unsigned long long)*n,
         info.data_type == 0 ? "ascii" : (info.data_type == 1 ? "binary" : "binary_compressed"),
         dense ? "dense" : "has non-finite points");
  return pts;
}

int main(int argc, char** argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: %s source.pcd target.pcd [out.pcd] [max_corr_dist] [max_iterations]\n", argv[0]);
    return 2;
  }
  uint64_t ns = 0, nt = 0;
  float* src = load_cloud(argv[1], &ns);
  float* tgt = load_cloud(argv[2], &nt);
  if (!src || !tgt) return 1;

  pclhip_ctx* ctx = NULL;
  CHECK(pclhip_ctx_create(0, NULL, &ctx), NULL);
  pclhip_index* index = NULL;
  CHECK(pclhip_index_build(ctx, tgt, 16, nt, NULL, 0, &index), ctx);
  const float viewpoint[3] = {0.0f, 0.0f, 0.0f};
  uint64_t nan_normals = 0;
  CHECK(pclhip_normals(index, 10, viewpoint, NULL, 0, &nan_normals), ctx);
  printf("target indexed in %.2f ms, normals (k = 10) in %.2f ms, %llu undefined\n", pclhip_index_build_ms(index),
         pclhip_index_last_kernel_ms(index), (unsigned long long)nan_normals);

  pclhip_icp* icp = NULL;
  CHECK(pclhip_icp_create(index, &icp), ctx);
  CHECK(pclhip_icp_set_source(icp, src, 16, ns), ctx);
  pclhip_icp_params p;
  pclhip_icp_params_default(&p);
  p.mode = PCLHIP_ICP_POINT_TO_PLANE;
  p.max_iterations = argc > 5 ? atoi(argv[5]) : 50;
  p.max_correspondence_distance = argc > 4 ? atof(argv[4]) : 0.05;
  p.transformation_epsilon = 1e-8;
  pclhip_icp_result r;
  CHECK(pclhip_icp_align(icp, &p, NULL, &r), ctx);
  double score = DBL_MAX;
  uint64_t used = 0;
  CHECK(pclhip_icp_fitness_score(icp, r.final_transformation, DBL_MAX, &score, &used), ctx);
  printf("converged %d after %d iterations (state %d), %llu correspondences, fitness %.6g, GPU %.3f ms\n", r.converged,
         r.nr_iterations, r.convergence_state, (unsigned long long)r.num_correspondences, score, r.gpu_ms);
  for (int i = 0; i < 4; ++i)
    printf("  %10.6f %10.6f %10.6f %10.6f\n", r.final_transformation[4 * i], r.final_transformation[4 * i + 1],
           r.final_transformation[4 * i + 2], r.final_transformation[4 * i + 3]);
  if (argc > 3) {
    float* out = (float*)malloc((size_t)(ns ? ns : 1) * 16);
    memcpy(out, src, (size_t)ns * 16);
    CHECK(pclhip_transform_cloud(ctx, r.final_transformation, 0, src, out, 16, ns, 0), ctx);
    CHECK(pclhip_pcd_write(argv[3], out, 16, 0, ns, 2, 8), ctx);
    printf("wrote %s\n", argv[3]);
    free(out);
  }
  pclhip_icp_destroy(icp);
  pclhip_index_destroy(index);
  pclhip_ctx_destroy(ctx);
  free(src);
  free(tgt);
  return 0;
}
