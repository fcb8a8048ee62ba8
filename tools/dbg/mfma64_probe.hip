// layout probe of v_mfma_f64_16x16x4_f64: A = e_i e_k^T style one-hot inputs, prints which (row, col) each lane/reg holds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe(double *out)
{
    const int lane = threadIdx.x;
    // A[i][k] = 100 i + k + 1 (i = lane & 15, k = lane >> 4 assumed); B[k][j] = one-hot test later
    // test 1: B = identity-like: B[k][j] = (j == k) -> D[i][j] = A[i][j] for j < 4
    double a = 100.0 * (lane & 15) + (lane >> 4) + 1;
    double b = ((lane & 15) == (lane >> 4)) ? 1.0 : 0.0;
    d4 c = {0, 0, 0, 0};
    d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = d[r];
}
int main()
{
    double *o; hipMalloc(&o, 256 * 8);
    probe<<<1, 64>>>(o);
    double h[256]; hipMemcpy(h, o, sizeof h, hipMemcpyDeviceToHost);
    // expected if A lane=(i=lane&15,k=lane>>4), B lane=(j=lane&15,k=lane>>4): D[i][j] = A[i][j] = 100 i + j + 1 for j<4 else 0
    int ok1 = 1, ok2 = 1;
    for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 4; ++r) {
            double v = h[lane * 4 + r];
            { int col = lane & 15, row = (lane >> 4) + 4 * r; double e = col < 4 ? 100.0 * row + col + 1 : 0; if (v != e) ok1 = 0; }
            { int col = lane & 15, row = 4 * (lane >> 4) + r; double e = col < 4 ? 100.0 * row + col + 1 : 0; if (v != e) ok2 = 0; }
        }
    printf("layout row=(lane>>4)+4*reg: %d   layout row=4*(lane>>4)+reg: %d\n", ok1, ok2);
    for (int lane = 0; lane < 64; lane += 5) printf("lane %2d: %g %g %g %g\n", lane, h[lane*4], h[lane*4+1], h[lane*4+2], h[lane*4+3]);
    return 0;
}
