// xm_dense_la.hip — inverse of a symmetric positive definite matrix on the device (blocked Cholesky + two blocked triangular
// solves with the identity), float64.  Used once per matrix-free context for the reduced camera Laplacian VT of
// utils/creatematrix.py:150-166 (the reference solves with it on the host through scipy.linalg.solve, :260); it is set-up work,
// O(n^3): the GEMM-shaped updates (all but O(n^2 b) of the flops) run on the f64 matrix cores, the 64-wide panel kernels are plain
// VALU code.
//
//   A = L L^T            right-looking: potrf on the 64 x 64 diagonal block, panel solve A21 <- A21 L11^-T, trailing update
//                        A22 -= A21 A21^T (lower tiles only)
//   L Y = I, L^T X = Y   block forward / backward substitution, the off-diagonal work as one GEMM per block row
// All matrices column-major with leading dimension ld; the result is the full symmetric inverse.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "xm_solver.h"

namespace xm {

constexpr int kLaB = 64;   // block size

// C[m x n] -= opA(A)[m x k] * opB(B)[k x n];  ta / tb: 0 = as stored, 1 = transposed.  64 x 64 tile per workgroup, K in chunks of 16
// through LDS, the products on the f64 matrix cores: each of the four wavefronts owns a 32 x 32 quarter as 2 x 2 blocks of
// v_mfma_f64_16x16x4_f64.  The MFMA computes the TRANSPOSED block (its "A" operand is taken from the B tile, its "B" operand from the
// A tile), so that the 16 lanes that share an accumulator register hold 16 consecutive ROWS of one column of C: C is column-major
// and the read-modify-write of the result is coalesced.  Operand / result maps of the f64 form (cdna_hip_programming.md): A[i = lane %
// 16][k = lane / 16], B[k = lane / 16][j = lane % 16], D[row = lane / 16 + 4 reg][col = lane % 16].
// lower_only: skip tiles strictly above the diagonal (syrk).
typedef double la_v4 __attribute__((ext_vector_type(4)));
// The K chunks are DOUBLE-BUFFERED (round 4): chunk k + 1 travels from memory into registers while the matrix cores work on chunk k out
// of LDS, and is stored into the other LDS buffer behind the one barrier per chunk.  The substitutions of the inverse launch at most
// n / 64 workgroups with K up to n: one workgroup per CU, so nothing else hides the load latency (13 681 rows: 0.8 s -> see
// profiles/r04_schur_setup.txt).
__global__ __launch_bounds__(256) void la_gemm_sub_kernel(int m, int n, int k, const double *__restrict__ A, int64_t lda, int ta,
                                                           const double *__restrict__ B, int64_t ldb, int tb, double *__restrict__ C,
                                                           int64_t ldc, int lower_only) {
    const int bi = blockIdx.x, bj = blockIdx.y;
    if (lower_only && bj > bi) return;
    __shared__ double As[2][16][kLaB + 1], Bs[2][16][kLaB + 1];   // [buffer][kk][i / j]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = bi * kLaB, j0 = bj * kLaB;
    const int r0 = (wave >> 1) * 32, c0 = (wave & 1) * 32;   // this wavefront's quarter of the tile
    const int l16 = lane & 15, lk = lane >> 4;
    la_v4 acc[2][2];   // [column block][row block] of the quarter, transposed blocks (see above)
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = la_v4{0.0, 0.0, 0.0, 0.0};
    // each thread moves 4 elements of either operand per chunk; consecutive threads run along the contiguous direction of the stored matrix
    int ia[4], ka[4], jb[4], kb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = threadIdx.x + 256 * q;
        ia[q] = ta ? e / 16 : e % kLaB; ka[q] = ta ? e % 16 : e / kLaB;
        jb[q] = tb ? e % kLaB : e / 16; kb[q] = tb ? e / kLaB : e % 16;
    }
    double va[4], vb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            va[q] = 0.0; vb[q] = 0.0;
            if (i0 + ia[q] < m && k0 + ka[q] < k)
                va[q] = ta ? A[(size_t)(k0 + ka[q]) + (size_t)(i0 + ia[q]) * lda] : A[(size_t)(i0 + ia[q]) + (size_t)(k0 + ka[q]) * lda];
            if (j0 + jb[q] < n && k0 + kb[q] < k)
                vb[q] = tb ? B[(size_t)(j0 + jb[q]) + (size_t)(k0 + kb[q]) * ldb] : B[(size_t)(k0 + kb[q]) + (size_t)(j0 + jb[q]) * ldb];
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { As[buf][ka[q]][ia[q]] = va[q]; Bs[buf][kb[q]][jb[q]] = vb[q]; }
    };
    fetch(0);
    stash(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < k; k0 += 16) {
        const bool more = k0 + 16 < k;
        if (more) fetch(k0 + 16);
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
            const int kk = kq * 4 + lk;
            double av[2], bv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) { av[u] = As[buf][kk][r0 + 16 * u + l16]; bv[u] = Bs[buf][kk][c0 + 16 * u + l16]; }
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[x], av[y], acc[x][y], 0, 0, 0);
        }
        if (more) stash(buf ^ 1);   // the other buffer: nobody reads it during this chunk
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gi = i0 + r0 + 16 * y + l16, gj = j0 + c0 + 16 * x + lk + 4 * r;   // D row <-> column of C, D column <-> row of C
                if (gi < m && gj < n) C[(size_t)gi + (size_t)gj * ldc] -= acc[x][y][r];
            }
}

// Cholesky of one b x b diagonal block (b <= 64) in place (lower), one workgroup; *info = 1 when a pivot is not positive
__global__ __launch_bounds__(256) void la_potrf_kernel(int b, double *__restrict__ A, int64_t lda, int *info) {
    __shared__ double L[kLaB][kLaB + 1];
    for (int e = threadIdx.x; e < b * b; e += 256) L[e % b][e / b] = A[(size_t)(e % b) + (size_t)(e / b) * lda];
    __syncthreads();
    for (int j = 0; j < b; ++j) {
        if (threadIdx.x == 0) {
            const double d = L[j][j];
            if (!(d > 0.0)) *info = 1;
            L[j][j] = sqrt(d > 0.0 ? d : 1.0);
        }
        __syncthreads();
        const double dj = L[j][j];
        for (int i = j + 1 + threadIdx.x; i < b; i += 256) L[i][j] /= dj;
        __syncthreads();
        for (int e = threadIdx.x; e < (b - j - 1) * (b - j - 1); e += 256) {   // trailing update of the lower part
            const int i = j + 1 + e % (b - j - 1), c = j + 1 + e / (b - j - 1);
            if (i >= c) L[i][c] -= L[i][j] * L[c][j];
        }
        __syncthreads();
    }
    for (int e = threadIdx.x; e < b * b; e += 256) {
        const int i = e % b, c = e / b;
        A[(size_t)i + (size_t)c * lda] = (i >= c) ? L[i][c] : 0.0;
    }
}

// rows of P (m x b) <- P * L^-T  with L the b x b lower factor: one thread per row (forward substitution along the row)
__global__ __launch_bounds__(256) void la_trsm_right_kernel(int m, int b, const double *__restrict__ Lm, int64_t ldl, double *__restrict__ P,
                                                             int64_t ldp) {
    __shared__ double L[kLaB][kLaB + 1];
    for (int e = threadIdx.x; e < b * b; e += 256) L[e % b][e / b] = Lm[(size_t)(e % b) + (size_t)(e / b) * ldl];
    __syncthreads();
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= m) return;
    double x[kLaB];
    for (int c = 0; c < b; ++c) {
        double t = P[(size_t)r + (size_t)c * ldp];
        for (int q = 0; q < c; ++q) t -= x[q] * L[c][q];
        x[c] = t / L[c][c];
    }
    for (int c = 0; c < b; ++c) P[(size_t)r + (size_t)c * ldp] = x[c];
}

// columns of T (b x n) <- L^-1 T (trans = 0) or L^-T T (trans = 1): one thread per column
__global__ __launch_bounds__(256) void la_trsm_left_kernel(int b, int n, const double *__restrict__ Lm, int64_t ldl, int trans,
                                                            double *__restrict__ T, int64_t ldt) {
    __shared__ double L[kLaB][kLaB + 1];
    for (int e = threadIdx.x; e < b * b; e += 256) L[e % b][e / b] = Lm[(size_t)(e % b) + (size_t)(e / b) * ldl];
    __syncthreads();
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    double x[kLaB];
    double *col = T + (size_t)c * ldt;
    if (!trans) {
        for (int i = 0; i < b; ++i) {
            double t = col[i];
            for (int q = 0; q < i; ++q) t -= L[i][q] * x[q];
            x[i] = t / L[i][i];
        }
    } else {
        for (int i = b - 1; i >= 0; --i) {
            double t = col[i];
            for (int q = i + 1; q < b; ++q) t -= L[q][i] * x[q];
            x[i] = t / L[i][i];
        }
    }
    for (int i = 0; i < b; ++i) col[i] = x[i];
}

__global__ __launch_bounds__(256) void la_identity_kernel(int n, double *X, int64_t ld) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)n * n) return;
    X[(size_t)(e % n) + (size_t)(e / n) * ld] = (e % n == e / n) ? 1.0 : 0.0;
}

// A (n x n, column-major, ld = n) -= q * u u^T : the contribution of one hub landmark to the reduced camera Laplacian (a landmark seen
// by every camera is a full rank-1 update of the (N-1)^2 matrix: 187 M entries at N = 13 682, seconds on the host, half a
// millisecond here)
__global__ __launch_bounds__(256) void la_rank1_sub_kernel(int64_t n, double *__restrict__ A, const double *__restrict__ u, double q) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * n) return;
    const int64_t r = e % n, c = e / n;
    const double ur = u[r], uc = u[c];
    if (ur != 0.0 && uc != 0.0) A[e] -= ur * uc * q;
}
void rank1_sub_device(int n, double *A, const double *u, double q, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(la_rank1_sub_kernel, dim3((unsigned)(((int64_t)n * n + 255) / 256)), dim3(256), 0, st, (int64_t)n, A, u, q);
    check_launch("rank1_sub");
}

static void gemm_sub(int m, int n, int k, const double *A, int64_t lda, int ta, const double *B, int64_t ldb, int tb, double *C, int64_t ldc,
                     int lower_only, hipStream_t st) {
    if (m <= 0 || n <= 0 || k <= 0) return;
    const dim3 g((m + kLaB - 1) / kLaB, (n + kLaB - 1) / kLaB);
    hipLaunchKernelGGL(la_gemm_sub_kernel, g, dim3(256), 0, st, m, n, k, A, lda, ta, B, ldb, tb, C, ldc, lower_only);
}

// A (device, column-major n x n, ld = n, lower triangle read) is overwritten by its Cholesky factor; X (device, n x n) receives the LOWER
// triangle of A^-1 (rows >= columns; what lies above the diagonal is scratch): Y = L^-1 is lower triangular, and of X = L^-T Y only the
// lower triangle is computed -- both substitutions then touch the columns [0, i + b) of block row i only, half the flops of the full
// solves with the identity.  spd_inverse_layout() writes the full symmetric matrix in the dense kernel's layout from it.
// Returns false when A is not positive definite.
bool spd_inverse_device(int n, double *A, double *X, hipStream_t st, bool trace) {
    if (n <= 0) return true;
    const int64_t ld = n;
    DevBuf<int> info;
    info.alloc(1);
    for (int k = 0; k < n; k += kLaB) {
        const int b = std::min(kLaB, n - k), m = n - k - b;
        double *A11 = A + (size_t)k + (size_t)k * ld;
        hipLaunchKernelGGL(la_potrf_kernel, dim3(1), dim3(256), 0, st, b, A11, ld, info.p);
        if (m > 0) {
            double *A21 = A11 + b;
            hipLaunchKernelGGL(la_trsm_right_kernel, dim3((m + 255) / 256), dim3(256), 0, st, m, b, A11, ld, A21, ld);
            gemm_sub(m, m, b, A21, ld, 0, A21, ld, 1, A11 + b + (size_t)b * ld, ld, 1, st);   // A22 -= A21 A21^T (lower tiles)
        }
    }
    check_launch("spd_inverse(cholesky)");
    auto t0 = std::chrono::steady_clock::now();
    int h = 0;
    XM_HIP_CHECK(hipMemcpyAsync(&h, info.p, sizeof(int), hipMemcpyDeviceToHost, st));
    XM_HIP_CHECK(hipStreamSynchronize(st));
    if (h) return false;
    if (trace) { std::fprintf(stderr, "spd_inverse: Cholesky factorisation %8.1f ms\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3); t0 = std::chrono::steady_clock::now(); }
    hipLaunchKernelGGL(la_identity_kernel, dim3((unsigned)(((int64_t)n * n + 255) / 256)), dim3(256), 0, st, n, X, ld);
    // Both substitutions advance in SUPER blocks of kLaS = 256 rows: one GEMM with everything outside the super block (K up to n, 4 x
    // n / 64 workgroups: every CU busy), then its four 64-row blocks in turn (small GEMM inside the super block + the 64 x 64 triangular
    // solve).  With 64-row steps alone the chain was 2 x 214 launches of at most 214 workgroups each.
    constexpr int kLaS = 4 * kLaB;
    for (int I = 0; I < n; I += kLaS) {   // forward: Y_I = L_II^-1 (I_I - L[I, 0:I] Y[0:I, :]); Y is lower triangular: columns [0, row end)
        const int B = std::min(kLaS, n - I);
        gemm_sub(B, I + B, I, A + I, ld, 0, X, ld, 0, X + I, ld, 0, st);
        for (int i = I; i < I + B; i += kLaB) {
            const int b = std::min(kLaB, n - i), nc = i + b;
            gemm_sub(b, nc, i - I, A + (size_t)i + (size_t)I * ld, ld, 0, X + I, ld, 0, X + i, ld, 0, st);
            hipLaunchKernelGGL(la_trsm_left_kernel, dim3((nc + 255) / 256), dim3(256), 0, st, b, nc, A + (size_t)i + (size_t)i * ld, ld, 0, X + i, ld);
        }
    }
    for (int I = ((n - 1) / kLaS) * kLaS; I >= 0; I -= kLaS) {   // backward: X_I = L_II^-T (Y_I - L[I+B:, I]^T X[I+B:, :]), lower triangle: columns [0, I + B)
        const int B = std::min(kLaS, n - I), below = n - I - B, ncI = I + B;
        gemm_sub(B, ncI, below, A + (size_t)(I + B) + (size_t)I * ld, ld, 1, X + I + B, ld, 0, X + I, ld, 0, st);
        for (int i = I + ((B - 1) / kLaB) * kLaB; i >= I; i -= kLaB) {
            const int b = std::min(kLaB, n - i), inside = I + B - i - b;   // rows of the super block below block i
            // (all the columns of the super block's range: the entries right of block i's own diagonal are scratch, but the blocks above need
            // X[i, c] for c up to their own diagonal only, which lies left of i + b)
            gemm_sub(b, i + b, inside, A + (size_t)(i + b) + (size_t)i * ld, ld, 1, X + i + b, ld, 0, X + i, ld, 0, st);
            hipLaunchKernelGGL(la_trsm_left_kernel, dim3((i + b + 255) / 256), dim3(256), 0, st, b, i + b, A + (size_t)i + (size_t)i * ld, ld, 1, X + i, ld);
        }
    }
    check_launch("spd_inverse(solve)");
    XM_HIP_CHECK(hipStreamSynchronize(st));
    if (trace) std::fprintf(stderr, "spd_inverse: two triangular substitutions (lower triangle) %8.1f ms\n",
                            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3);
    return true;
}

// dst (row-major, n rows, leading dimension ldd, zero padding untouched) = the full symmetric matrix whose LOWER triangle X holds (column-major, ld = n)
__global__ __launch_bounds__(256) void la_sym_layout_kernel(int64_t n, const double *__restrict__ X, double *__restrict__ dst, int64_t ldd) {
    __shared__ double T[64][65];
    // tile (bi, bj) of dst, bj >= bi: upper tiles come transposed from the mirror tile of X's lower triangle, lower tiles straight
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    if (c0 > r0) {          // strictly upper tile of dst: dst[r][c] = X(c, r), read coalesced along X's rows (index c) and transposed through LDS
        for (int k = ty; k < 64; k += 4) {
            const int64_t c = c0 + tx, r = r0 + k;
            T[k][tx] = (c < n && r < n) ? X[(size_t)c + (size_t)r * n] : 0.0;   // X(c, r): row c >= column r
        }
        __syncthreads();
        for (int k = ty; k < 64; k += 4) {
            const int64_t r = r0 + k, c = c0 + tx;
            if (r < n && c < n) dst[(size_t)r * ldd + c] = T[k][tx];
        }
    } else {                // diagonal and lower tiles: dst[r][c] = X(max(r, c), min(r, c)); X is column-major, so read along r and transpose
        for (int k = ty; k < 64; k += 4) {
            const int64_t r = r0 + tx, c = c0 + k;
            double v = 0.0;
            if (r < n && c < n) v = (r >= c) ? X[(size_t)r + (size_t)c * n] : X[(size_t)c + (size_t)r * n];
            T[tx][k] = v;
        }
        __syncthreads();
        for (int k = ty; k < 64; k += 4) {
            const int64_t r = r0 + k, c = c0 + tx;
            if (r < n && c < n) dst[(size_t)r * ldd + c] = T[k][tx];
        }
    }
}
void spd_inverse_layout(int n, const double *X, double *dst, int64_t ldd, hipStream_t st) {
    if (n <= 0) return;
    const unsigned nt = (unsigned)((n + 63) / 64);
    hipLaunchKernelGGL(la_sym_layout_kernel, dim3(nt, nt), dim3(256), 0, st, (int64_t)n, X, dst, ldd);
    check_launch("spd_inverse_layout");
}

}  // namespace xm
