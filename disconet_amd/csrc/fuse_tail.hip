// K5 (layers 2-4) + K6: per-pixel pairwise attention tail, softmax over the
// agents of a scene and weighted sum of the (warped) neighbour maps.
//
// Layer 1 of the attention MLP (1x1 conv 2C -> 128 on cat[ego, nbr]) is linear
// in its input, so it is split W1 = [W1_ego | W1_nbr] and evaluated on the
// MFMA conv engine once per map instead of once per pair:
//     E      = ego . W1_ego^T + b1          (shared by every neighbour of the ego)
//     F_self = ego . W1_nbr^T,  F_j = warp(j->i) . W1_nbr^T
// This kernel finishes the MLP per (ego, neighbour, pixel): BN1+ReLU on E+F,
// 128->32->8->1 with folded bias/BN + ReLU, exp / sum over neighbours (no
// max-shift, as the reference) and the weighted sum over the C channels.
// HBM/L2-bound: reads A maps once per ego, writes one fused map.
//
// One workgroup = 8 pixels of one (sample, ego): thread (p, o) = (tid>>5, tid&31)
// owns hidden unit o of pixel p in layer 2; the weighted sum maps tid -> channel
// so every pixel row is one coalesced C-float access.
//
// Replaces PixelWeightedFusionSoftmax.forward (layers 2-4) and the fusion loop
// body of upstream:coperception/models/det/DiscoNet.py :: DiscoNet.forward
// (Appx A.5).
#include "dn_internal.h"

namespace {

constexpr int PIX = 8;        // pixels per workgroup
constexpr int H1 = 128, H2 = 32, H3 = 8;
constexpr int MAX_NBR = 8;    // ego + up to 7 neighbours

struct TailArgs {
  const float* feat; const float* warped; const float* g; const float* fw;
  const int32_t* num_agent;
  dn_mlp_tail_params p;
  int batch, agents, hw, c, only_v2i, ego_first, ego_count;
  float* fused; float* weights_out;
};

__global__ void __launch_bounds__(256)
disco_fuse_tail_kernel(const TailArgs a) {
  __shared__ float w2s[H2][H1 + 1];
  __shared__ float w3s[H3][H2 + 1];
  __shared__ float h1s[PIX][H1];
  __shared__ float h2s[PIX][H2 + 1];
  __shared__ float h3s[PIX][H3 + 1];
  __shared__ float es[MAX_NBR][PIX];      // exp(s_k) per neighbour / pixel
  __shared__ int nbr_of[MAX_NBR];         // neighbour agent index per slot k

  const int tid = threadIdx.x;
  const int bi = blockIdx.y;                             // b * ego_count + (i - ego_first)
  const int b = bi / a.ego_count, il = bi % a.ego_count, i = a.ego_first + il;
  const int p0 = blockIdx.x * PIX;
  int n_live = a.num_agent[b];
  n_live = n_live < 0 ? 0 : (n_live > a.agents ? a.agents : n_live);   // a bad count never indexes past the agents
  const int img = il * a.batch + b;                      // agent-major index among LOCAL egos
  const float* ego = a.feat + ((size_t)i * a.batch + b) * a.hw * a.c;   // feat holds all agents
  float* out = a.fused + (size_t)img * a.hw * a.c;

  if (i >= n_live) {
    // padded agent: its map passes through un-fused
    // wave-uniform trip counts with a per-lane `if` inside (a round-1 convention; DESIGN.md 3.6 (B) shows the
    // loop form was not what mattered)
    const int n_it = (PIX * a.c + 255) / 256;
    for (int it = 0; it < n_it; ++it) {
      const int idx = tid + 256 * it;
      const int p = p0 + idx / a.c;
      if (idx < PIX * a.c && p < a.hw) out[(size_t)p * a.c + idx % a.c] = ego[(size_t)p * a.c + idx % a.c];
    }
    return;
  }

  // neighbour list: ego first, then j ascending (j != i), honouring only_v2i
  int nk = 0;
  if (tid == 0) {
    nbr_of[0] = i;
    int k = 1;
    for (int j = 0; j < n_live && k < MAX_NBR; ++j) {
      if (j == i) continue;
      if (a.only_v2i && i != 0 && j != 0) continue;
      nbr_of[k++] = j;
    }
    for (int r = k; r < MAX_NBR; ++r) nbr_of[r] = -1;
  }
  for (int idx = tid; idx < H2 * H1; idx += 256) w2s[idx / H1][idx % H1] = a.p.w2[idx];
  if (tid < H3 * H2) w3s[tid / H2][tid % H2] = a.p.w3[tid];
  __syncthreads();
  while (nk < MAX_NBR && nbr_of[nk] >= 0) ++nk;

  const int pl = tid >> 5, o = tid & 31;   // layer-2 role
  for (int k = 0; k < nk; ++k) {
    const int j = nbr_of[k];
    // ---- layer 1 finish: h1 = relu(bn1(E + F_k)), 8 pixels x 128 units
    for (int idx = tid; idx < PIX * H1; idx += 256) {
      const int p = idx / H1, u = idx % H1;
      const int pix = p0 + p;
      float v = 0.f;
      if (pix < a.hw) {
        const float e = a.g[((size_t)img * a.hw + pix) * (2 * H1) + u];
        float f;
        if (k == 0) {
          f = a.g[((size_t)img * a.hw + pix) * (2 * H1) + H1 + u];
        } else {
          const int jj = j - (j > i ? 1 : 0);
          f = a.fw[(((size_t)bi * (a.agents - 1) + jj) * a.hw + pix) * H1 + u];
        }
        v = fmaxf((e + f) * a.p.bn1_scale[u] + a.p.bn1_shift[u], 0.f);
      }
      h1s[p][u] = v;
    }
    __syncthreads();
    // ---- layer 2: 128 -> 32
    {
      float acc = 0.f;
#pragma unroll 8
      for (int u = 0; u < H1; ++u) acc += w2s[o][u] * h1s[pl][u];
      h2s[pl][o] = fmaxf(acc * a.p.s2[o] + a.p.t2[o], 0.f);
    }
    __syncthreads();
    // ---- layer 3: 32 -> 8
    if (tid < PIX * H3) {
      const int p = tid >> 3, q = tid & 7;
      float acc = 0.f;
#pragma unroll
      for (int u = 0; u < H2; ++u) acc += w3s[q][u] * h2s[p][u];
      h3s[p][q] = fmaxf(acc * a.p.s3[q] + a.p.t3[q], 0.f);
    }
    __syncthreads();
    // ---- layer 4: 8 -> 1, ReLU, exp
    if (tid < PIX) {
      float acc = 0.f;
#pragma unroll
      for (int u = 0; u < H3; ++u) acc += a.p.w4[u] * h3s[tid][u];
      const float s = fmaxf(acc + a.p.b4[0], 0.f);
      es[k][tid] = expf(s);
    }
    __syncthreads();
  }

  // ---- softmax over neighbours (sum in list order, as the reference)
  if (tid < PIX) {
    float sum = 0.f;
    for (int k = 0; k < nk; ++k) sum += es[k][tid];
    for (int k = 0; k < nk; ++k) {
      const float wgt = es[k][tid] / sum;
      es[k][tid] = wgt;
      if (a.weights_out && p0 + tid < a.hw)
        a.weights_out[((size_t)bi * a.agents + k) * a.hw + p0 + tid] = wgt;
    }
  }
  __syncthreads();

  // ---- fused = sum_k w_k * nbr_k ; thread -> channel
  for (int p = 0; p < PIX; ++p) {
    const int pix = p0 + p;
    if (pix >= a.hw) break;
    for (int it = 0; it < (a.c + 255) / 256; ++it) {
      const int ch = tid + 256 * it;
      if (ch >= a.c) continue;
      float acc = es[0][p] * ego[(size_t)pix * a.c + ch];
      for (int k = 1; k < nk; ++k) {
        const int j = nbr_of[k];
        const int jj = j - (j > i ? 1 : 0);
        acc += es[k][p] *
               a.warped[(((size_t)bi * (a.agents - 1) + jj) * a.hw + pix) * a.c + ch];
      }
      out[(size_t)pix * a.c + ch] = acc;
    }
  }
}

}  // namespace

extern "C" int dn_disco_fuse_tail(const float* feat, const float* warped, const float* g,
                                  const float* fw, const int32_t* num_agent,
                                  const dn_mlp_tail_params* p, int batch, int agents, int hw,
                                  int c, int only_v2i, int ego_first, int ego_count,
                                  float* fused, float* weights_out, void* stream) {
  DN_REQUIRE(feat && g && num_agent && p && fused, "fuse_tail: null pointer");
  DN_REQUIRE(agents < 2 || (warped && fw), "fuse_tail: neighbours present but warped/fw null");
  DN_REQUIRE(batch > 0 && agents > 0 && hw > 0 && c > 0, "fuse_tail: empty problem");
  DN_REQUIRE(ego_first >= 0 && ego_count > 0 && ego_first + ego_count <= agents,
             "fuse_tail: ego range [%d, %d) outside 0..%d", ego_first, ego_first + ego_count, agents);
  DN_REQUIRE(agents <= MAX_NBR, "fuse_tail: at most %d agents supported (got %d)", MAX_NBR, agents);
  DN_REQUIRE(p->bn1_scale && p->bn1_shift && p->w2 && p->s2 && p->t2 && p->w3 && p->s3 &&
                 p->t3 && p->w4 && p->b4, "fuse_tail: null MLP parameter");
  TailArgs a;
  a.feat = feat; a.warped = warped; a.g = g; a.fw = fw; a.num_agent = num_agent; a.p = *p;
  a.batch = batch; a.agents = agents; a.hw = hw; a.c = c; a.only_v2i = only_v2i;
  a.ego_first = ego_first; a.ego_count = ego_count;
  a.fused = fused; a.weights_out = weights_out;
  dim3 grid((hw + PIX - 1) / PIX, batch * ego_count);
  hipLaunchKernelGGL(disco_fuse_tail_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
  return dn::check_launch("disco_fuse_tail_kernel");
}
