// Small assembly kernels of the Pfaffian (BCS / Nambu) path.
//
//  nambu_assemble_kernel  Bogoliubov matrix of a cut side from the mode blocks found on the device:
//                         column j of the result = M2C( [conj] src[:, col_src[j]] ) -- the column order
//                         [a modes | a^dag modes], the conjugate halves and the Majorana -> complex
//                         fermion row transformation of pfaffian.py:880-895 (`nambu`, `vector_M2C`).
//  nambu_w_kernel         W = [[ Vr[L:, L:], Q ], [ P, 0 ]] with Q = [ I[:, idx1] | Vr[L:, idx2] ],
//                         P = [ Vr[idx1, L:] ; I[idx2, :] ]: the LU/Schur kernel then returns
//                         det(U^*) (Onishi norm, pfaffian.py:1352-1359) and -P (U^*)^-1 Q, whose blocks
//                         are AA, BA, BB of pfaffian.py:1384-1391.
//  pf_matrix_kernel       N = [[BB, BA], [-BA^T, AA]] with AA, BB antisymmetrised (pfaffian.py:1394-1400).
#include "common.hpp"

namespace tmf {

__global__ __launch_bounds__(256) void nambu_assemble_kernel(const tmf_nambu_asm_desc* __restrict__ desc) {
  const tmf_nambu_asm_desc d = desc[blockIdx.x];
  const cd* __restrict__ src = reinterpret_cast<const cd*>(d.src);
  cd* __restrict__ dst = reinterpret_cast<cd*>(d.dst);
  const int32_t* cs = reinterpret_cast<const int32_t*>(d.col_src);
  const int8_t* cj = reinterpret_cast<const int8_t*>(d.col_conj);
  const int n = d.n2 / 2;  // sites
  const double r2 = 0.70710678118654752440;
  for (int e = threadIdx.x; e < n * d.n2; e += 256) {
    const int x = e % n, j = e / n;
    const int sc_ = cs[j];
    cd v0 = src[(size_t)(2 * x) + (size_t)sc_ * d.lds_];
    cd v1 = src[(size_t)(2 * x + 1) + (size_t)sc_ * d.lds_];
    if (cj[j]) {
      v0.y = -v0.y;
      v1.y = -v1.y;
    }
    // M2C (pfaffian.py:120-124):  out0 = (v0 - i v1)/sqrt2 ;  out1 = (v0 + i v1)/sqrt2
    const cd o0 = make_cd((v0.x + v1.y) * r2, (v0.y - v1.x) * r2);
    const cd o1 = make_cd((v0.x - v1.y) * r2, (v0.y + v1.x) * r2);
    dst[(size_t)(2 * x) + (size_t)j * d.ldd] = o0;
    dst[(size_t)(2 * x + 1) + (size_t)j * d.ldd] = o1;
  }
}

__global__ __launch_bounds__(256) void nambu_w_kernel(const tmf_nambu_w_desc* __restrict__ desc) {
  const tmf_nambu_w_desc d = desc[blockIdx.x];
  const cd* __restrict__ Vr = reinterpret_cast<const cd*>(d.Vr);
  cd* __restrict__ W = reinterpret_cast<cd*>(d.W);
  const int32_t* i1 = reinterpret_cast<const int32_t*>(d.idx1);
  const int32_t* i2 = reinterpret_cast<const int32_t*>(d.idx2);
  const int L = d.L, a = d.na, b = d.nb, m = L + a + b;
  for (int e = threadIdx.x; e < m * m; e += 256) {
    const int r = e % m, c = e / m;
    cd v = make_cd(0.0, 0.0);
    if (r < L && c < L) {
      v = Vr[(size_t)(L + r) + (size_t)(L + c) * d.ldv];          // U^* block
    } else if (r < L && c < L + a) {
      v = make_cd(r == i1[c - L] ? 1.0 : 0.0, 0.0);               // I[:, idx1]
    } else if (r < L) {
      v = Vr[(size_t)(L + r) + (size_t)i2[c - L - a] * d.ldv];    // Vr[L:, idx2]
    } else if (c < L && r < L + a) {
      v = Vr[(size_t)i1[r - L] + (size_t)(L + c) * d.ldv];        // Vr[idx1, L:]
    } else if (c < L) {
      v = make_cd(c == i2[r - L - a] ? 1.0 : 0.0, 0.0);           // I[idx2, :]
    }
    W[(size_t)r + (size_t)c * d.ldw] = v;
  }
}

__global__ __launch_bounds__(256) void pf_matrix_kernel(const tmf_pf_matrix_desc* __restrict__ desc) {
  const tmf_pf_matrix_desc d = desc[blockIdx.x];
  const cd* __restrict__ S = reinterpret_cast<const cd*>(d.S);   // S = -P (U^*)^-1 Q, (a+b) x (a+b)
  cd* __restrict__ N = reinterpret_cast<cd*>(d.N);
  const int a = d.na, b = d.nb, m = a + b;
  // S rows/cols: [idx1 part (a) | idx2 part (b)]:  -S[:a,:a] = AA, -S[a:,:a] = BA, -S[a:,a:] = BB
  auto AA = [&](int i, int j) { const cd s = S[(size_t)i + (size_t)j * d.lds_]; return make_cd(-s.x, -s.y); };
  auto BA = [&](int i, int j) { const cd s = S[(size_t)(a + i) + (size_t)j * d.lds_]; return make_cd(-s.x, -s.y); };
  auto BB = [&](int i, int j) { const cd s = S[(size_t)(a + i) + (size_t)(a + j) * d.lds_]; return make_cd(-s.x, -s.y); };
  for (int e = threadIdx.x; e < m * m; e += 256) {
    const int r = e % m, c = e / m;
    cd v;
    if (r < b && c < b) {
      const cd x = BB(r, c), y = BB(c, r);
      v = make_cd(0.5 * (x.x - y.x), 0.5 * (x.y - y.y));
    } else if (r < b) {
      v = BA(r, c - b);
    } else if (c < b) {
      const cd x = BA(c, r - b);
      v = make_cd(-x.x, -x.y);
    } else {
      const cd x = AA(r - b, c - b), y = AA(c - b, r - b);
      v = make_cd(0.5 * (x.x - y.x), 0.5 * (x.y - y.y));
    }
    N[(size_t)r + (size_t)c * d.ldn] = v;
  }
}

}  // namespace tmf

using namespace tmf;

namespace tmf {
// Onishi norm sqrt(prod sv(U)) = |det U|^(1/2) of every site (pfaffian.py:1352-1359) from the determinants the LU left
__global__ __launch_bounds__(256) void onishi_norm_kernel(const cd* __restrict__ det, cd* __restrict__ out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = make_cd(sqrt(sqrt(fma(det[i].x, det[i].x, det[i].y * det[i].y))), 0.0);
}
}  // namespace tmf

extern "C" int tmf_onishi_norms(const void* d_det, void* d_norm, int n, void* stream) {
  if (n <= 0) return TMF_OK;
  hipLaunchKernelGGL(tmf::onishi_norm_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const tmf::cd*>(d_det), static_cast<tmf::cd*>(d_norm), n);
  return tmf::check_hip(hipGetLastError(), "tmf_onishi_norms launch");
}

extern "C" int tmf_nambu_assemble_batched(const tmf_nambu_asm_desc* d_desc, int nprob, void* stream) {
  if (nprob <= 0) return TMF_OK;
  hipLaunchKernelGGL(nambu_assemble_kernel, dim3(nprob), dim3(256), 0, static_cast<hipStream_t>(stream), d_desc);
  return check_hip(hipGetLastError(), "tmf_nambu_assemble_batched");
}

extern "C" int tmf_nambu_w_batched(const tmf_nambu_w_desc* d_desc, int nprob, void* stream) {
  if (nprob <= 0) return TMF_OK;
  hipLaunchKernelGGL(nambu_w_kernel, dim3(nprob), dim3(256), 0, static_cast<hipStream_t>(stream), d_desc);
  return check_hip(hipGetLastError(), "tmf_nambu_w_batched");
}

extern "C" int tmf_pf_matrix_batched(const tmf_pf_matrix_desc* d_desc, int nprob, void* stream) {
  if (nprob <= 0) return TMF_OK;
  hipLaunchKernelGGL(pf_matrix_kernel, dim3(nprob), dim3(256), 0, static_cast<hipStream_t>(stream), d_desc);
  return check_hip(hipGetLastError(), "tmf_pf_matrix_batched");
}
