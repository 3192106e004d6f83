// K5: explicit inverse of a symmetric positive-definite matrix, entirely on the device.
//
// The reference inverts the Schur complement of the least-squares prox with Eigen::LDLT +
// solve(Identity) on one CPU thread (reference src/epsilon/linear/dense_matrix_impl.cc:21-30)
// and then applies the explicit inverse by dgemv every iteration.  Same contract here (the
// cached operator is an explicit inverse, because a GEMV is the GPU-friendly apply), built as
//
//   1. blocked right-looking Cholesky  W = L L^T        (64-wide panels; the panel solve and the
//      trailing update are GEMMs on the MFMA kernel, the 64x64 diagonal block is factored
//      and inverted inside one workgroup's LDS),
//   2. X = L^-1 by recursive doubling: inv([L11 0; L21 L22]) = [X11 0; -X22 L21 X11, X22],
//      bottom-up from the 64x64 diagonal inverses -- every step is two GEMMs,
//   3. W^-1 = X^T X  (lower tiles, then mirrored).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "kernels.h"

namespace eps {
namespace k {

namespace {

constexpr int NB = 64;
constexpr int kBlock = 256;

// Factor the kb x kb block at W (ld) in place (lower Cholesky) and write inv(L) (dense NB x NB,
// zeros above the diagonal, ld = NB) to Dinv.  One workgroup.  *flag != 0 on a bad pivot.
template <class T>
__global__ __launch_bounds__(kBlock) void PotrfDiagKernel(T* W, int64_t ld, int kb, T* Dinv,
                                                          int* flag) {
  __shared__ T S[NB][NB + 1];
  __shared__ T col[NB];
  const int t = threadIdx.x;
  for (int idx = t; idx < NB * NB; idx += kBlock) {
    const int r = idx % NB, c = idx / NB;
    T v = T(0);
    if (r < kb && c < kb) v = (r >= c) ? W[r + c * ld] : T(0);
    S[r][c] = v;
  }
  __syncthreads();
  for (int j = 0; j < kb; ++j) {
    if (t == 0) {
      T d = S[j][j];
      if (!(d > T(0))) {
        *flag = 1;
        d = T(1);
      }
      S[j][j] = sqrt(d);
    }
    __syncthreads();
    const T djj = S[j][j];
    for (int i = j + 1 + t; i < kb; i += kBlock) S[i][j] /= djj;
    __syncthreads();
    const int w = kb - j - 1;
    for (int idx = t; idx < w * w; idx += kBlock) {
      const int r = j + 1 + idx % w, c = j + 1 + idx / w;
      if (r >= c) S[r][c] -= S[r][j] * S[c][j];
    }
    __syncthreads();
  }
  // the factor goes back to W before S is overwritten by its inverse
  for (int idx = t; idx < NB * NB; idx += kBlock) {
    const int r = idx % NB, c = idx / NB;
    if (r < kb && c < kb && r >= c) W[r + c * ld] = S[r][c];
  }
  __syncthreads();
  // in-place inverse of the lower-triangular factor, last column first (trti2 order):
  //   X[j][j] = 1/L[j][j] ;  X[j+1:, j] = -X[j+1:, j+1:] * L[j+1:, j] * X[j][j]
  for (int j = kb - 1; j >= 0; --j) {
    if (t < NB) col[t] = (t > j && t < kb) ? S[t][j] : T(0);
    __syncthreads();
    const T ajj = T(1) / S[j][j];
    T nv = T(0);
    const bool mine = (t > j && t < kb);
    if (mine) {
      T acc = T(0);
      for (int kk = j + 1; kk <= t; ++kk) acc += S[t][kk] * col[kk];
      nv = -acc * ajj;
    }
    __syncthreads();
    if (mine) S[t][j] = nv;
    if (t == j) S[j][j] = ajj;
    __syncthreads();
  }
  for (int idx = t; idx < NB * NB; idx += kBlock) {
    const int r = idx % NB, c = idx / NB;
    Dinv[r + c * NB] = (r < kb && c < kb && r >= c) ? S[r][c] : T(0);
  }
}

DVec Sub(const DVec& W, int64_t i, int64_t j, int64_t ld) {
  const int64_t off = i + j * ld;
  return W.Slice(off, W.n - off);
}

}  // namespace

void SpdInverseInPlace(const DVec& W, int64_t n) {
  EPS_CHECK(W.n >= n * n);
  if (n == 0) return;
  Runtime& rt = Runtime::Get();
  hipStream_t s = rt.stream();
  ProfScope prof("spd_inverse", n);
  const DType dt = W.dt;
  const int64_t ld = n;
  const int64_t nb = (n + NB - 1) / NB;

  DVec dinv = DVec::Empty(nb * NB * NB, dt);
  auto flag_buf = rt.Alloc(sizeof(int));
  int* flag = static_cast<int*>(flag_buf->p);
  EPS_HIP(hipMemsetAsync(flag, 0, sizeof(int), s));
  DVec panel = DVec::Empty(std::max<int64_t>(n, 1) * NB, dt);

  // ---- 1. blocked Cholesky ------------------------------------------------------------------
  for (int64_t kblk = 0; kblk < nb; ++kblk) {
    const int64_t k0 = kblk * NB;
    const int kb = static_cast<int>(std::min<int64_t>(NB, n - k0));
    DVec Wkk = Sub(W, k0, k0, ld);
    DVec Dk = dinv.Slice(kblk * NB * NB, NB * NB);
    if (dt == F32) {
      hipLaunchKernelGGL(PotrfDiagKernel<float>, dim3(1), dim3(kBlock), 0, s, Wkk.as<float>(), ld,
                         kb, Dk.as<float>(), flag);
    } else {
      hipLaunchKernelGGL(PotrfDiagKernel<double>, dim3(1), dim3(kBlock), 0, s, Wkk.as<double>(),
                         ld, kb, Dk.as<double>(), flag);
    }
    const int64_t rem = n - (k0 + kb);
    if (rem <= 0) continue;
    DVec W21 = Sub(W, k0 + kb, k0, ld);
    DVec tmp = panel.Slice(0, rem * kb);
    MatCopy(false, rem, kb, 1.0, W21, ld, tmp);
    // L21 = W21 * inv(L11)^T
    Gemm(false, true, rem, kb, kb, 1.0, tmp, rem, Dk, NB, 0.0, W21, ld);
    // W22 -= L21 L21^T  (lower tiles only)
    DVec W22 = Sub(W, k0 + kb, k0 + kb, ld);
    Gemm(false, true, rem, rem, kb, -1.0, W21, ld, W21, ld, 1.0, W22, ld, true);
  }

  // ---- 2. X = inv(L) by recursive doubling ---------------------------------------------------
  DVec X = DVec::Zeros(n * n, dt);
  for (int64_t kblk = 0; kblk < nb; ++kblk) {
    const int64_t k0 = kblk * NB;
    const int64_t kb = std::min<int64_t>(NB, n - k0);
    // copy the kb x kb inverse block (ld NB) into X's diagonal block (ld n)
    EPS_HIP(hipMemcpy2DAsync(Sub(X, k0, k0, ld).data(), ld * DTypeSize(dt),
                             dinv.Slice(kblk * NB * NB, NB * NB).data(), NB * DTypeSize(dt),
                             kb * DTypeSize(dt), kb, hipMemcpyDeviceToDevice, s));
  }
  DVec tmp2 = DVec::Empty(std::max<int64_t>(1, (n / 2 + NB) * (n / 2 + NB)), dt);
  for (int64_t sz = NB; sz < n; sz *= 2) {
    for (int64_t r0 = 0; r0 + sz < n; r0 += 2 * sz) {
      const int64_t s1 = sz;                                  // rows/cols of block 1
      const int64_t s2 = std::min<int64_t>(sz, n - (r0 + sz));  // rows of block 2
      DVec L21 = Sub(W, r0 + s1, r0, ld);
      DVec X11 = Sub(X, r0, r0, ld);
      DVec X22 = Sub(X, r0 + s1, r0 + s1, ld);
      DVec X21 = Sub(X, r0 + s1, r0, ld);
      EPS_CHECK(tmp2.n >= s2 * s1);
      DVec T = tmp2.Slice(0, s2 * s1);
      Gemm(false, false, s2, s1, s1, 1.0, L21, ld, X11, ld, 0.0, T, s2);
      Gemm(false, false, s2, s1, s2, -1.0, X22, ld, T, s2, 0.0, X21, ld);
    }
  }

  // ---- 3. W^-1 = X^T X ------------------------------------------------------------------------
  Gemm(true, false, n, n, n, 1.0, X, ld, X, ld, 0.0, W, ld, true);
  SymmetrizeFromLower(W, n, ld);

  int host_flag = 0;
  EPS_HIP(hipMemcpyAsync(&host_flag, flag, sizeof(int), hipMemcpyDeviceToHost, s));
  EPS_HIP(hipStreamSynchronize(s));
  EPS_CHECK_MSG(host_flag == 0, "dense inverse: matrix is not positive definite");
}

}  // namespace k
}  // namespace eps
