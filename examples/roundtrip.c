/* roundtrip.c -- the C ABI from a plain C host (no Python, no HIP headers): wpdall + bestbasistree(JBB) +
 * iwpdall by that tree on host arrays, i.e. what the Julia shim does through ccall.
 *   gcc -std=c99 -Iinclude examples/roundtrip.c -o roundtrip -Lwaveletsext.jl_amd/csrc -lwaveletsext_hip -lm
 *   LD_LIBRARY_PATH=waveletsext.jl_amd/csrc ./roundtrip
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "waveletsext_hip.h"

#define CHECK(call)                                                                          \
    do { int rc_ = (call); if (rc_ != WX_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, wx_last_error()); return 1; } } while (0)

int main(void)
{
    const int64_t n = 256, B = 37;
    const int L = 8;
    const double s2 = 0.70710678118654752440;
    const double qmf[2] = {s2, s2};                                   /* WT.qmf(wavelet(WT.haar)) */
    double *x = malloc(sizeof(double) * n * B), *xw = malloc(sizeof(double) * n * (L + 1) * B);
    double *xr = malloc(sizeof(double) * n * B), *sum = malloc(sizeof(double) * n * (L + 1));
    double *sq = malloc(sizeof(double) * n * (L + 1)), *costs = malloc(sizeof(double) * ((1 << (L + 1)) - 1));
    uint8_t *tree = malloc((size_t)(n - 1));
    unsigned long long st = 88172645463325252ull;
    for (int64_t i = 0; i < n * B; ++i) {                             /* xorshift noise on a ramp */
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        x[i] = (double)(i % n) / (double)n + ((double)(st >> 11) / 9007199254740992.0 - 0.5);
    }
    printf("library version %d, %d HIP device(s)\n", wx_version(), wx_device_count());
    CHECK(wx_wpd1d_f64(x, xw, n, L, B, qmf, 2, NULL));                                   /* wpdall(x, wt, L) */
    CHECK(wx_jbb_moments_f64(xw, sum, sq, n * (L + 1), B, 0, NULL));                     /* tree_costs(Xw, JBB()) ... */
    CHECK(wx_jbb_costs_f64(sum, sq, B, n, L + 1, 0, 0, 2.0, costs, NULL));
    CHECK(wx_treeselect_f64(costs, ((int64_t)1 << (L + 1)) - 1, n, 0, tree));            /* ... bestbasis_treeselection */
    int64_t kept = 0;
    for (int64_t i = 0; i < n - 1; ++i) kept += tree[i];
    CHECK(wx_iwpd1d_f64(xw, xr, n, L + 1, L, tree, n - 1, B, qmf, 2, NULL));             /* iwpdall(Xw, wt, tree) */
    double err = 0.0;
    for (int64_t i = 0; i < n * B; ++i) { const double d = fabs(xr[i] - x[i]); if (d > err) err = d; }
    printf("best-basis tree keeps %lld of %lld nodes; round trip max error %.3e\n", (long long)kept, (long long)(n - 1), err);
    CHECK(wx_shutdown());
    free(x); free(xw); free(xr); free(sum); free(sq); free(costs); free(tree);
    return err < 1e-10 ? 0 : 2;
}
