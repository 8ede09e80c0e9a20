/*
 * CPU oracle (TEST INFRASTRUCTURE ONLY) for the selective scan the reference calls at
 * mlagg/nnunetv2/training/nnUNetTrainer/variants/mamba/MambaSkip.py:445-451
 * (mamba-ssm `selective_scan_fn`, third-party and absent from /root/reference: semantics
 * restated from the published recurrence -- parity unpinned at this boundary).
 *
 *   delta' = softplus(delta + delta_bias)            (softplus threshold 20, as torch)
 *   h_l    = exp(delta'_l A[d][n]) h_{l-1} + delta'_l B[b][g][n][l] u[b][d][l]
 *   y_l    = sum_n C[b][g][n][l] h_l[n] + D[d] u[b][d][l]            g = d / (dim / groups)
 *
 * All arithmetic in double, results rounded to float once: this is the "truth" the HIP
 * kernels and the torch restatement are both compared with.  Built by oracle/Makefile into
 * oracle/_build/libmlagg_oracle.so (gcc -O2 -fopenmp); never linked by the product.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static double softplus_d(double x) { return x > 20.0 ? x : log1p(exp(x)); }
static double sigmoid_d(double x) { return 1.0 / (1.0 + exp(-x)); }

int oracle_selscan_fwd(const float *u, const float *delta, const float *A, const float *Bm, const float *Cm,
                       const float *Dv, const float *dbias, float *out, int batch, int dim, int L, int N,
                       int G, int softplus)
{
    const int H = dim / G;
#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int b = 0; b < batch; ++b)
        for (int d = 0; d < dim; ++d) {
            const int g = d / H;
            const float *ur = u + ((size_t)b * dim + d) * L;
            const float *dr = delta + ((size_t)b * dim + d) * L;
            const float *Br = Bm + ((size_t)b * G + g) * N * L;
            const float *Cr = Cm + ((size_t)b * G + g) * N * L;
            float *yr = out + ((size_t)b * dim + d) * L;
            double h[64];
            for (int n = 0; n < N; ++n) h[n] = 0.0;
            for (int l = 0; l < L; ++l) {
                double dl = (double)dr[l] + (dbias ? (double)dbias[d] : 0.0);
                if (softplus) dl = softplus_d(dl);
                const double uu = ur[l];
                double y = 0.0;
                for (int n = 0; n < N; ++n) {
                    h[n] = exp(dl * (double)A[d * N + n]) * h[n] + dl * (double)Br[(size_t)n * L + l] * uu;
                    y += (double)Cr[(size_t)n * L + l] * h[n];
                }
                yr[l] = (float)(y + (Dv ? (double)Dv[d] * uu : 0.0));
            }
        }
    return 0;
}

/* Gradients of the above w.r.t. every input.  dB/dC are summed over the channels of a
 * group, dA/dD/ddelta_bias over batch and time, exactly as autograd does for the
 * broadcasted operands in the eager restatement. */
int oracle_selscan_bwd(const float *u, const float *delta, const float *A, const float *Bm, const float *Cm,
                       const float *Dv, const float *dbias, const float *dout, float *du, float *ddelta,
                       float *dA, float *dB, float *dC, float *dD, float *ddbias, int batch, int dim, int L,
                       int N, int G, int softplus)
{
    const int H = dim / G;
    double *dA_acc = (double *)calloc((size_t)dim * N, sizeof(double));
    double *dD_acc = (double *)calloc(dim, sizeof(double));
    double *db_acc = (double *)calloc(dim, sizeof(double));
    if (!dA_acc || !dD_acc || !db_acc) return 1;
#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int b = 0; b < batch; ++b)
        for (int g = 0; g < G; ++g) {
            const float *Br = Bm + ((size_t)b * G + g) * N * L;
            const float *Cr = Cm + ((size_t)b * G + g) * N * L;
            double *dBg = (double *)calloc((size_t)N * L, sizeof(double));
            double *dCg = (double *)calloc((size_t)N * L, sizeof(double));
            double *hs = (double *)malloc((size_t)(L + 1) * N * sizeof(double));
            double *dl = (double *)malloc((size_t)L * sizeof(double));
            for (int c = 0; c < H; ++c) {
                const int d = g * H + c;
                const float *ur = u + ((size_t)b * dim + d) * L;
                const float *dr = delta + ((size_t)b * dim + d) * L;
                const float *gr = dout + ((size_t)b * dim + d) * L;
                float *dur = du + ((size_t)b * dim + d) * L;
                float *ddr = ddelta + ((size_t)b * dim + d) * L;
                for (int n = 0; n < N; ++n) hs[n] = 0.0;
                for (int l = 0; l < L; ++l) {
                    double x = (double)dr[l] + (dbias ? (double)dbias[d] : 0.0);
                    dl[l] = softplus ? softplus_d(x) : x;
                    for (int n = 0; n < N; ++n)
                        hs[(size_t)(l + 1) * N + n] = exp(dl[l] * (double)A[d * N + n]) * hs[(size_t)l * N + n] +
                                                      dl[l] * (double)Br[(size_t)n * L + l] * (double)ur[l];
                }
                double q[64];           /* q[n] = a_{l+1} * (dL/dh_{l+1}) flowing into h_l */
                double dAl[64];
                for (int n = 0; n < N; ++n) { q[n] = 0.0; dAl[n] = 0.0; }
                double dDl = 0.0, dbl = 0.0;
                for (int l = L - 1; l >= 0; --l) {
                    const double gy = gr[l], uu = ur[l];
                    double dd = 0.0, duu = (Dv ? (double)Dv[d] : 0.0) * gy;
                    dDl += gy * uu;
                    for (int n = 0; n < N; ++n) {
                        const double An = A[d * N + n];
                        const double a = exp(dl[l] * An);
                        const double Bv = Br[(size_t)n * L + l], Cv = Cr[(size_t)n * L + l];
                        const double gh = q[n] + gy * Cv;                 /* dL/dh_l */
                        dCg[(size_t)n * L + l] += gy * hs[(size_t)(l + 1) * N + n];
                        const double da = gh * hs[(size_t)l * N + n];     /* dL/da_l */
                        dd += da * a * An + gh * Bv * uu;
                        dAl[n] += da * a * dl[l];
                        dBg[(size_t)n * L + l] += gh * dl[l] * uu;
                        duu += gh * dl[l] * Bv;
                        q[n] = a * gh;
                    }
                    if (softplus) {
                        double x = (double)dr[l] + (dbias ? (double)dbias[d] : 0.0);
                        dd *= (x > 20.0 ? 1.0 : sigmoid_d(x));
                    }
                    dbl += dd;
                    dur[l] = (float)duu;
                    ddr[l] = (float)dd;
                }
#pragma omp critical
                {
                    for (int n = 0; n < N; ++n) dA_acc[d * N + n] += dAl[n];
                    dD_acc[d] += dDl;
                    db_acc[d] += dbl;
                }
            }
            float *dBo = dB + ((size_t)b * G + g) * N * L;
            float *dCo = dC + ((size_t)b * G + g) * N * L;
            for (size_t i = 0; i < (size_t)N * L; ++i) { dBo[i] = (float)dBg[i]; dCo[i] = (float)dCg[i]; }
            free(dBg); free(dCg); free(hs); free(dl);
        }
    for (int i = 0; i < dim * N; ++i) dA[i] = (float)dA_acc[i];
    for (int i = 0; i < dim; ++i) { if (dD) dD[i] = (float)dD_acc[i]; if (ddbias) ddbias[i] = (float)db_acc[i]; }
    free(dA_acc); free(dD_acc); free(db_acc);
    return 0;
}
