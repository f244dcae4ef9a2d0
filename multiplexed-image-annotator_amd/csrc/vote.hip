// Per-cell vote over one or two models' softmax outputs (reference cell_type_annotation/model.py:481-633,
// tie-break order = key order of utils.py:143-146 get_void_vote).  One thread per cell; all compares in fp32, which is
// what the reference's numpy-scalar arithmetic does (np.float32 probabilities against weak Python-float thresholds).
//
// Global class ids: 0..16 = void-vote key order, 17 = "Others".
//   two models : vote[g] = p (class sets are disjoint apart from Others); winner = first maximum in id order;
//                thresh = type_conf[winner] < 0 ? min(o1, o2, conf) : type_conf[winner];
//                vote < thresh -> (Others, -1) else (winner, vote)
//   one model  : winner = first maximum over the model's own class order (Others included);
//                thresh = type_conf[winner] > 0 ? type_conf[winner] : conf;
//                winner != Others && p < thresh -> (Others, -1) else (winner, p)
#include "ribca_common.h"
#include "ribca_kernels.h"

namespace ribca {

constexpr int kOthers = 17;

__global__ __launch_bounds__(256) void vote_kernel(VoteArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  int8_t lab;
  float cf;
  if (a.p_b != nullptr) {
    float vote[kOthers];
#pragma unroll
    for (int g = 0; g < kOthers; ++g) vote[g] = 0.f;
    float o1 = 0.f, o2 = 0.f;
    for (int k = 0; k < a.k_a; ++k) {
      const float p = a.p_a[(size_t)i * a.k_a + k];
      const int g = a.map_a[k];
      if (g == kOthers) o1 = p;
      else {
#pragma unroll
        for (int t = 0; t < kOthers; ++t) if (t == g) vote[t] += p;
      }
    }
    for (int k = 0; k < a.k_b; ++k) {
      const float p = a.p_b[(size_t)i * a.k_b + k];
      const int g = a.map_b[k];
      if (g == kOthers) o2 = p;
      else {
#pragma unroll
        for (int t = 0; t < kOthers; ++t) if (t == g) vote[t] += p;
      }
    }
    int best = 0;
    float bv = vote[0];
#pragma unroll
    for (int g = 1; g < kOthers; ++g) if (vote[g] > bv) { bv = vote[g]; best = g; }
    const float tc = a.type_conf[best];
    // Python min(o1, o2, conf): keeps the first of equal values; value-equal in fp32 either way
    const float thresh = tc < 0.f ? fminf(fminf(o1, o2), a.conf) : tc;
    if (bv < thresh) { lab = kOthers; cf = -1.f; }
    else { lab = (int8_t)best; cf = bv; }
  } else {
    int bk = 0;
    float bv = a.p_a[(size_t)i * a.k_a];
    for (int k = 1; k < a.k_a; ++k) {
      const float p = a.p_a[(size_t)i * a.k_a + k];
      if (p > bv) { bv = p; bk = k; }
    }
    const int g = a.map_a[bk];
    const float tc = a.type_conf[g];
    const float thresh = tc > 0.f ? tc : a.conf;
    if (g != kOthers && bv < thresh) { lab = kOthers; cf = -1.f; }
    else { lab = (int8_t)g; cf = bv; }
  }
  a.label[i] = lab;
  a.out_conf[i] = cf;
}

void launch_vote(const VoteArgs& a, hipStream_t s) {
  if (a.n <= 0) return;
  hipLaunchKernelGGL(vote_kernel, dim3((a.n + 255) / 256), dim3(256), 0, s, a);
}

}  // namespace ribca
