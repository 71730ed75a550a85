/* ORACLE — TEST INFRASTRUCTURE ONLY (never linked into libvqhip.so).
 *
 * VQ codebook nearest-neighbour lookup.  The reference repository contains NO quantizer
 * (grep for quantiz/codebook/argmin/vq finds nothing; ae.py:336-348 is a DiagonalGaussian
 * with std 0.00), so there is nothing in the reference to restate or to pin against: PARITY UNPINNED
 * with respect to the reference.  What this file is pinned to instead (tests/test_vq_oracle_pin.py, CPU): the mathematical
 * definition argmin_j sum_k (z_ik - e_jk)^2 by fp64 brute force and the PUBLISHED VQGAN quantizer arithmetic (taming-transformers
 * VectorQuantizer: |z|^2 + |e|^2 - 2 z.e^T, torch.argmin) — exact agreement on well-separated data, lowest index on exact
 * ties, every choice an fp64 near-minimiser on adversarial near-ties (disagreement rates between fp32 orders are reported).
 * This file DEFINES the evaluation order the HIP kernel (csrc/optim_vq.hip) must reproduce bit-for-bit:
 *
 *   zz_i  = fma-chain over k ascending, start 0:  zz = fmaf(z_ik, z_ik, zz)
 *   ee_j  = same over the code vector
 *   dot   = same:                                 dot = fmaf(z_ik, e_jk, dot)
 *   d_ij  = (zz_i - 2*dot) + ee_j                 (2*dot is exact; two roundings)
 *   idx_i = argmin_j d_ij, scanning j ascending with strict '<'  (lowest index wins ties,
 *           torch.argmin's CPU behaviour)
 * i.e. the standard VQGAN distance |z|^2 - 2 z.e + |e|^2 evaluated in fp32 with one fixed
 * association order, so that near-ties resolve identically on every implementation.
 * Build: gcc -O2 -std=c99 -ffp-contract=off (the explicit fmaf calls are the only fused ops).
 */
#include <math.h>
#include <stdint.h>

void vq_nearest_oracle(const float* z, const float* cb, int64_t n_tokens, int n_codes, int dim, int64_t* idx,
                       float* min_dist) {
  for (int64_t i = 0; i < n_tokens; ++i) {
    const float* zi = z + i * dim;
    float zz = 0.f;
    for (int k = 0; k < dim; ++k) zz = fmaf(zi[k], zi[k], zz);
    float best = INFINITY;
    int64_t bi = 0;
    for (int j = 0; j < n_codes; ++j) {
      const float* e = cb + (int64_t)j * dim;
      float ee = 0.f, dot = 0.f;
      for (int k = 0; k < dim; ++k) ee = fmaf(e[k], e[k], ee);
      for (int k = 0; k < dim; ++k) dot = fmaf(zi[k], e[k], dot);
      const float d = (zz - 2.f * dot) + ee;
      if (d < best) { best = d; bi = j; }
    }
    idx[i] = bi;
    if (min_dist) min_dist[i] = best;
  }
}
