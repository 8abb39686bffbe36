/* CPU oracle for the VQ codebook lookup.  TEST INFRASTRUCTURE (see oracle/__init__.py).
 *
 * Restates taming/modules/vqvae/quantize.py:306-310 of the reference:
 *     d = sum(z^2, dim=1) + sum(e^2, dim=1) - 2 * z @ e^T ;  idx = argmin(d, dim=1)
 * in plain fp32 C with a FIXED operation order, so that the HIP kernel can be bit-exact
 * against it:   zz  = fmaf chain over k ascending of z[k]*z[k]        (start 0)
 *               ee  = fmaf chain over k ascending of e[k]*e[k]        (start 0)
 *               dot = fmaf chain over k ascending of z[k]*e[k]        (start 0)
 *               d   = (zz + ee) - 2*dot        (two roundings; 2*dot is exact)
 *               argmin = first (lowest) index of the minimum, as torch.argmin.
 * The reference evaluates the same expression with torch reductions / an MKL GEMM whose
 * summation order is unspecified; the two agree wherever the best and second-best
 * distances differ by more than a few ulp (tests check exactly that on the goldens).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (see oracle/build.py).
 */
#include <math.h>
#include <stdint.h>

void oracle_vq_sqnorm(const float *e, int64_t n, int64_t dim, float *ee) {
    for (int64_t j = 0; j < n; ++j) {
        float s = 0.0f;
        for (int64_t k = 0; k < dim; ++k) s = fmaf(e[j * dim + k], e[j * dim + k], s);
        ee[j] = s;
    }
}

/* z [rows, dim], e [n, dim] -> idx [rows] (int64), optional dmin [rows] */
void oracle_vq_argmin(const float *z, const float *e, int64_t rows, int64_t n, int64_t dim,
                      int64_t *idx, float *dmin) {
    float ee_stack[4096];
    float *ee = ee_stack; /* n <= 4096 in every configuration of the path */
    oracle_vq_sqnorm(e, n, dim, ee);
    for (int64_t r = 0; r < rows; ++r) {
        const float *zr = z + r * dim;
        float zz = 0.0f;
        for (int64_t k = 0; k < dim; ++k) zz = fmaf(zr[k], zr[k], zz);
        float best = INFINITY;
        int64_t bi = 0;
        for (int64_t j = 0; j < n; ++j) {
            const float *ej = e + j * dim;
            float dot = 0.0f;
            for (int64_t k = 0; k < dim; ++k) dot = fmaf(zr[k], ej[k], dot);
            float d = (zz + ee[j]) - 2.0f * dot;
            if (d < best) { best = d; bi = j; }
        }
        idx[r] = bi;
        if (dmin) dmin[r] = best;
    }
}
