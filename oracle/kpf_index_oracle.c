/* ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Plain-C restatement of the two integer-valued sub-ops of the KeypointFusion forward, independent of torch:
 *   ball query   pointnet2_ops==3.0.0 ball_query (requirements.txt:15; call sites model/model.py:158,174) — third-party CUDA
 *                extension absent from /root/reference, restated from its published semantics ("parity unpinned")
 *   top-4 pixels dataloader/loader.py:958-963 (squared distances to every feature pixel, torch.topk(4, largest=False))
 * Only tests/ may load this library; it is checked against oracle/kpf_oracle.py (itself pinned to the imported reference) by
 * tests/test_index_oracle.py and then serves as a second, torch-free checker for the HIP kernels' index outputs.
 * Build: make -C oracle   (gcc, -ffp-contract=off so that every product and sum is rounded individually like the references).
 */
#include <stdint.h>

/* idx[s][0..nsample): the first `nsample` point indices n (ascending) with |new_xyz[s] - xyz[n]|^2 < radius^2; unfilled slots
 * repeat the first hit; all zero when there is none.  d^2 = dx*dx + dy*dy + dz*dz in fp32, left to right. */
void kpf_oracle_ball_query(const float* xyz, int N, const float* new_xyz, int S, float radius, int nsample, int32_t* idx) {
  const float r2 = radius * radius;
  for (int s = 0; s < S; ++s) {
    int32_t* out = idx + (long)s * nsample;
    int cnt = 0;
    for (int k = 0; k < nsample; ++k) out[k] = 0;
    for (int n = 0; n < N && cnt < nsample; ++n) {
      const float dx = new_xyz[s * 3 + 0] - xyz[n * 3 + 0];
      const float dy = new_xyz[s * 3 + 1] - xyz[n * 3 + 1];
      const float dz = new_xyz[s * 3 + 2] - xyz[n * 3 + 2];
      const float d2 = dx * dx + dy * dy + dz * dz;
      if (d2 < r2) {
        if (cnt == 0)
          for (int k = 0; k < nsample; ++k) out[k] = n;
        out[cnt++] = n;
      }
    }
  }
}

/* For each of N points the 4 pixels p (of P) with the smallest d^2 = (x-X)^2 + (y-Y)^2 + (z-Z)^2, ascending; ties keep the
 * lower pixel index first.  d2out may be NULL. */
void kpf_oracle_top4(const float* pcl, int N, const float* img_xyz, int P, int32_t* idx, float* d2out) {
  for (int n = 0; n < N; ++n) {
    float bd[4] = {3.4e38f, 3.4e38f, 3.4e38f, 3.4e38f};
    int32_t bi[4] = {-1, -1, -1, -1};
    for (int p = 0; p < P; ++p) {
      const float dx = pcl[n * 3 + 0] - img_xyz[p * 3 + 0];
      const float dy = pcl[n * 3 + 1] - img_xyz[p * 3 + 1];
      const float dz = pcl[n * 3 + 2] - img_xyz[p * 3 + 2];
      const float d2 = dx * dx + dy * dy + dz * dz;
      if (d2 < bd[3]) {
        int k = 3;
        while (k > 0 && d2 < bd[k - 1]) {
          bd[k] = bd[k - 1];
          bi[k] = bi[k - 1];
          --k;
        }
        bd[k] = d2;
        bi[k] = p;
      }
    }
    for (int k = 0; k < 4; ++k) {
      idx[n * 4 + k] = bi[k];
      if (d2out) d2out[n * 4 + k] = bd[k];
    }
  }
}
