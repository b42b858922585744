// Row-wise device kernels of the Mformer fusion (a1-a4, a9): everything between the GEMMs.
// One wave per entity (attention, read-outs) or per token row (LayerNorm); a row of D floats is
// spread over the wave as col = lane + 64*u, u < NU, so every load/store is a 256-byte segment.
#pragma once
#include "common.hip.h"

#include "small_mm.hip.h"

namespace snag {

template <int NU>
struct Cols {
    float v[NU];
};

template <int NU>
__device__ __forceinline__ Cols<NU> ld_cols(const float* __restrict__ p, int D, int lane) {
    Cols<NU> r;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int c = lane + 64 * u;
        r.v[u] = c < D ? p[c] : 0.f;
    }
    return r;
}

template <int NU>
__device__ __forceinline__ void st_cols(float* __restrict__ p, const Cols<NU>& r, int D, int lane) {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int c = lane + 64 * u;
        if (c < D) p[c] = r.v[u];
    }
}

// The H*M*M per-entity attention entries (keep flags, extra probability gradients) live in up to three registers per lane:
// entry e sits in lane e % 64 of register e / 64 (192 entries: the reference's default 5 heads with all 6 modalities is 180).
constexpr int kMaxAttnEntries = 192;
struct Entries3 {
    int a, b, c;
};
__device__ __forceinline__ Entries3 load_keep3(const uint8_t* __restrict__ keep, int64_t base, int HMM, int lane) {
    Entries3 r{1, 1, 1};
    if (keep != nullptr) {
        r.a = lane < HMM ? (int)keep[base + lane] : 1;
        r.b = lane + 64 < HMM ? (int)keep[base + 64 + lane] : 1;
        r.c = lane + 128 < HMM ? (int)keep[base + 128 + lane] : 1;
    }
    return r;
}
__device__ __forceinline__ Entries3 load_f32x3(const float* __restrict__ x, int64_t base, int HMM, int lane) {
    Entries3 r{0, 0, 0};
    if (x != nullptr) {
        r.a = lane < HMM ? __float_as_int(x[base + lane]) : 0;
        r.b = lane + 64 < HMM ? __float_as_int(x[base + 64 + lane]) : 0;
        r.c = lane + 128 < HMM ? __float_as_int(x[base + 128 + lane]) : 0;
    }
    return r;
}
// e is wave-uniform
__device__ __forceinline__ int entry_at(const Entries3& r, int e) {
    return e < 64 ? __builtin_amdgcn_readlane(r.a, e) : (e < 128 ? __builtin_amdgcn_readlane(r.b, e - 64) : __builtin_amdgcn_readlane(r.c, e - 128));
}

// ---------------------------------------------------------------------------------------------
// Self-attention over the M modality tokens of one entity.
// reference: BertSelfAttention.forward, model/SNAG_tools.py:178-209
//   S = Q K^T / sqrt(d) ; P = softmax(S) ; P~ = dropout(P) ; ctx = P~ V    (per head)
template <int M, int NU>
__device__ __forceinline__ void head_scores(const Cols<NU> (&q)[M], const Cols<NU> (&k)[M], int lo, int hi, int lane,
                                            float inv_sqrt_d, float (&p)[M][M]) {
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = 0; j < M; ++j) {
            float part = 0.f;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int c = lane + 64 * u;
                part += (c >= lo && c < hi) ? q[i].v[u] * k[j].v[u] : 0.f;
            }
            p[i][j] = sum64(part) * inv_sqrt_d;
        }
#pragma unroll
    for (int i = 0; i < M; ++i) {
        float mx = p[i][0];
#pragma unroll
        for (int j = 1; j < M; ++j) mx = fmaxf(mx, p[i][j]);
        float den = 0.f;
#pragma unroll
        for (int j = 0; j < M; ++j) {
            p[i][j] = expf(p[i][j] - mx);
            den += p[i][j];
        }
        const float inv = 1.0f / den;
#pragma unroll
        for (int j = 0; j < M; ++j) p[i][j] *= inv;
    }
}

template <int M, int NU>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ qkv, int N, int D, int H, float inv_sqrt_d,
                                                       const uint8_t* __restrict__ keep, float kscale,
                                                       float* __restrict__ probs, float* __restrict__ ctx) {
    const int64_t n = (int64_t)blockIdx.x * 4 + wave_id_uniform();
    if (n >= N) return;
    const int lane = threadIdx.x & 63;
    const int d = D / H;
    Cols<NU> q[M], k[M], v[M], o[M];
    // the entity's H*M*M keep flags in one coalesced read (lanes 0..63, 64..127), handed out by v_readlane below:
    // a per-(i,j) global load inside the head loop would be a dependent memory round trip each
    const int HMM = H * M * M;
    const Entries3 kb = load_keep3(keep, n * HMM, HMM, lane);
#pragma unroll
    for (int i = 0; i < M; ++i) {
        const float* row = qkv + (n * M + i) * (int64_t)(3 * D);
        q[i] = ld_cols<NU>(row, D, lane);
        k[i] = ld_cols<NU>(row + D, D, lane);
        v[i] = ld_cols<NU>(row + 2 * D, D, lane);
#pragma unroll
        for (int u = 0; u < NU; ++u) o[i].v[u] = 0.f;
    }
    for (int h = 0; h < H; ++h) {
        const int lo = h * d, hi = lo + d;
        float p[M][M];
        head_scores<M, NU>(q, k, lo, hi, lane, inv_sqrt_d, p);
        const int64_t pbase = (n * H + h) * (int64_t)(M * M);
        if (keep != nullptr) {
#pragma unroll
            for (int i = 0; i < M; ++i)
#pragma unroll
                for (int j = 0; j < M; ++j) {
                    const int e = h * M * M + i * M + j;
                    const int kf = entry_at(kb, e);
                    p[i][j] = kf ? p[i][j] * kscale : 0.f;
                }
        }
        // lanes 0..M*M-1 store one probability each (select chain keeps p[][] in registers)
        {
            float mine = 0.f;
#pragma unroll
            for (int i = 0; i < M; ++i)
#pragma unroll
                for (int j = 0; j < M; ++j) mine = (lane == i * M + j) ? p[i][j] : mine;
            if (lane < M * M) probs[pbase + lane] = mine;
        }
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int c = lane + 64 * u;
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < M; ++j) acc += p[i][j] * v[j].v[u];
                o[i].v[u] += (c >= lo && c < hi) ? acc : 0.f;
            }
    }
#pragma unroll
    for (int i = 0; i < M; ++i) st_cols<NU>(ctx + (n * M + i) * (int64_t)D, o[i], D, lane);
}

// backward of the above; P is recomputed from Q,K (a dropped probability still shapes the softmax
// Jacobian, so P~ alone is not enough).  dprobs_extra is the gradient that reaches P~ through the
// attention-derived modality weights (model/SNAG_tools.py:41-43), last layer only.
template <int M, int NU>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dctx, int N,
                                                       int D, int H, float inv_sqrt_d, const uint8_t* __restrict__ keep,
                                                       float kscale, const float* __restrict__ dprobs_extra,
                                                       float* __restrict__ dqkv, float* __restrict__ bias_part) {
    __shared__ float red[4][3][64 * NU];
    const int lane = threadIdx.x & 63;
    const int wave = wave_id_uniform();
    const int d = D / H;
    Cols<NU> sq, sk, sv;                 // running column sums of dQ, dK, dV over this wave's entities (bias gradient)
#pragma unroll
    for (int u = 0; u < NU; ++u) sq.v[u] = sk.v[u] = sv.v[u] = 0.f;
  for (int64_t n = (int64_t)blockIdx.x * 4 + wave; n < N; n += (int64_t)gridDim.x * 4) {
    Cols<NU> q[M], k[M], v[M], go[M], dq[M], dk[M], dv[M];
    const int HMM = H * M * M;
    const Entries3 kb = load_keep3(keep, n * HMM, HMM, lane);
    const Entries3 ex = load_f32x3(dprobs_extra, n * HMM, HMM, lane);
#pragma unroll
    for (int i = 0; i < M; ++i) {
        const float* row = qkv + (n * M + i) * (int64_t)(3 * D);
        q[i] = ld_cols<NU>(row, D, lane);
        k[i] = ld_cols<NU>(row + D, D, lane);
        v[i] = ld_cols<NU>(row + 2 * D, D, lane);
        go[i] = ld_cols<NU>(dctx + (n * M + i) * (int64_t)D, D, lane);
#pragma unroll
        for (int u = 0; u < NU; ++u) dq[i].v[u] = dk[i].v[u] = dv[i].v[u] = 0.f;
    }
    for (int h = 0; h < H; ++h) {
        const int lo = h * d, hi = lo + d;
        float p[M][M], ds[M][M];
        head_scores<M, NU>(q, k, lo, hi, lane, inv_sqrt_d, p);
#pragma unroll
        for (int i = 0; i < M; ++i) {
            float dp[M];
            float rowdot = 0.f;
#pragma unroll
            for (int j = 0; j < M; ++j) {
                float part = 0.f;
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int c = lane + 64 * u;
                    part += (c >= lo && c < hi) ? go[i].v[u] * v[j].v[u] : 0.f;
                }
                float g = sum64(part);                                   // d loss / d P~[i][j] through ctx
                const int e = h * M * M + i * M + j;
                if (dprobs_extra != nullptr)
                    g += __int_as_float(entry_at(ex, e));
                const int kf = entry_at(kb, e);
                const float km = (keep == nullptr) ? 1.f : (kf ? kscale : 0.f);
                // dV_j += P~[i][j] * dctx_i
                const float pt = p[i][j] * km;
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int c = lane + 64 * u;
                    dv[j].v[u] += (c >= lo && c < hi) ? pt * go[i].v[u] : 0.f;
                }
                dp[j] = g * km;                                          // d loss / d P[i][j]
                rowdot += dp[j] * p[i][j];
            }
#pragma unroll
            for (int j = 0; j < M; ++j) ds[i][j] = p[i][j] * (dp[j] - rowdot) * inv_sqrt_d;
        }
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int c = lane + 64 * u;
                const bool in = (c >= lo && c < hi);
                float aq = 0.f, ak = 0.f;
#pragma unroll
                for (int j = 0; j < M; ++j) {
                    aq += ds[i][j] * k[j].v[u];      // dQ_i = sum_j dS_ij K_j
                    ak += ds[j][i] * q[j].v[u];      // dK_i = sum_j dS_ji Q_j
                }
                dq[i].v[u] += in ? aq : 0.f;
                dk[i].v[u] += in ? ak : 0.f;
            }
    }
#pragma unroll
    for (int i = 0; i < M; ++i) {
        float* row = dqkv + (n * M + i) * (int64_t)(3 * D);
        st_cols<NU>(row, dq[i], D, lane);
        st_cols<NU>(row + D, dk[i], D, lane);
        st_cols<NU>(row + 2 * D, dv[i], D, lane);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            sq.v[u] += dq[i].v[u];
            sk.v[u] += dk[i].v[u];
            sv.v[u] += dv[i].v[u];
        }
    }
  }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        red[wave][0][lane + 64 * u] = sq.v[u];
        red[wave][1][lane + 64 * u] = sk.v[u];
        red[wave][2][lane + 64 * u] = sv.v[u];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * 64 * NU; i += 256) {
        const int which = i / (64 * NU), c = i % (64 * NU);
        if (c < D) bias_part[(int64_t)blockIdx.x * 3 * D + which * D + c] = ((red[0][which][c] + red[1][which][c]) + red[2][which][c]) + red[3][which][c];
    }
}

// ---------------------------------------------------------------------------------------------
// Read-out-only form of the two kernels above, for the LAST layer when only one token's output is consumed
// (joint_way Mformer_hd_graph reads hidden[:, g, :] alone): the query row, the attention row, the context and everything behind
// them exist for token g only; keys and values still come from all M tokens.  Same arithmetic per element as the full kernels
// (same summation order inside a row), so hidden[:, g, :] is the full computation's, and the gradients differ from it only by the
// summation order of the weight-gradient GEMMs (N rows instead of N*M rows of which 2/3 are exact zeros).
//   qkv [N*M, 3D] as above (the q part is valid in the rows of token g only), ctx_g [N, D] compact,
//   probs / keep [N,H,M,M] in the full layout (row g of each head read / written).
template <int M, int NU>
__device__ __forceinline__ void head_scores_row(const Cols<NU>& q, const Cols<NU> (&k)[M], int lo, int hi, int lane, float inv_sqrt_d,
                                                float (&p)[M]) {
#pragma unroll
    for (int j = 0; j < M; ++j) {
        float part = 0.f;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int c = lane + 64 * u;
            part += (c >= lo && c < hi) ? q.v[u] * k[j].v[u] : 0.f;
        }
        p[j] = sum64(part) * inv_sqrt_d;
    }
    float mx = p[0];
#pragma unroll
    for (int j = 1; j < M; ++j) mx = fmaxf(mx, p[j]);
    float den = 0.f;
#pragma unroll
    for (int j = 0; j < M; ++j) {
        p[j] = expf(p[j] - mx);
        den += p[j];
    }
    const float inv = 1.0f / den;
#pragma unroll
    for (int j = 0; j < M; ++j) p[j] *= inv;
}

template <int M, int NU>
__global__ __launch_bounds__(256) void attn_ro_fwd_kernel(const float* __restrict__ qkv, int N, int D, int H, int g, float inv_sqrt_d,
                                                          const uint8_t* __restrict__ keep, float kscale,
                                                          float* __restrict__ probs, float* __restrict__ ctx_g) {
    const int64_t n = (int64_t)blockIdx.x * 4 + wave_id_uniform();
    if (n >= N) return;
    const int lane = threadIdx.x & 63;
    const int d = D / H;
    const int HMM = H * M * M;
    const Entries3 kb = load_keep3(keep, n * HMM, HMM, lane);
    Cols<NU> k[M], v[M], o;
    const Cols<NU> q = ld_cols<NU>(qkv + (n * M + g) * (int64_t)(3 * D), D, lane);
#pragma unroll
    for (int i = 0; i < M; ++i) {
        const float* row = qkv + (n * M + i) * (int64_t)(3 * D);
        k[i] = ld_cols<NU>(row + D, D, lane);
        v[i] = ld_cols<NU>(row + 2 * D, D, lane);
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) o.v[u] = 0.f;
    for (int h = 0; h < H; ++h) {
        const int lo = h * d, hi = lo + d;
        float p[M];
        head_scores_row<M, NU>(q, k, lo, hi, lane, inv_sqrt_d, p);
        if (keep != nullptr) {
#pragma unroll
            for (int j = 0; j < M; ++j) {
                const int e = h * M * M + g * M + j;
                const int kf = entry_at(kb, e);
                p[j] = kf ? p[j] * kscale : 0.f;
            }
        }
        {
            float mine = 0.f;
#pragma unroll
            for (int j = 0; j < M; ++j) mine = (lane == j) ? p[j] : mine;
            if (lane < M) probs[(n * H + h) * (int64_t)(M * M) + g * M + lane] = mine;
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int c = lane + 64 * u;
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < M; ++j) acc += p[j] * v[j].v[u];
            o.v[u] += (c >= lo && c < hi) ? acc : 0.f;
        }
    }
    st_cols<NU>(ctx_g + n * (int64_t)D, o, D, lane);
}

//   dctx_g [N, D] -> dq_g [N, D] (compact) and dkv [N*M, 2D] (k then v); bias_part [block][3][D] = column sums of dq_g, dk, dv.
//   (ro3: see below)
// The attention backward of the read-out row: rows = what attn_ro_load fetched for an entity, go = d ctx_g [D] in the lanes (column lane + 64 u);
// writes dq / dk / dv and adds them to the wave's running bias sums.  ONE statement of the arithmetic for attn_ro_bwd_kernel and for the launch that
// forms d ctx itself (readout_fused.hip).
// an entity's rows as the attention backward reads them: the read-out token's query, every token's key and value, the keep flags
template <int M, int NU>
struct AttnRoRows {
    Cols<NU> q, k[M], v[M];
    Entries3 kb;
};
template <int M, int NU>
__device__ __forceinline__ AttnRoRows<M, NU> attn_ro_load(const float* __restrict__ qkv, int64_t n, int D, int H, int g, const uint8_t* __restrict__ keep, int lane) {
    AttnRoRows<M, NU> r;
    const int HMM = H * M * M;
    r.kb = load_keep3(keep, n * HMM, HMM, lane);
    r.q = ld_cols<NU>(qkv + (n * M + g) * (int64_t)(3 * D), D, lane);
#pragma unroll
    for (int i = 0; i < M; ++i) {
        const float* row = qkv + (n * M + i) * (int64_t)(3 * D);
        r.k[i] = ld_cols<NU>(row + D, D, lane);
        r.v[i] = ld_cols<NU>(row + 2 * D, D, lane);
    }
    return r;
}
// NE entities side by side (NE = 2 in the launch that forms d ctx itself: the two entities' dependent chains -- sum64 after sum64 -- interleave;
// every entity's own arithmetic and the order in which the entities enter the bias sums are those of NE = 1).  live[e] = 0: entity e is a
// stand-in (its rows were fetched at a clamped index): nothing of it is stored or summed.
template <int M, int NU, int NE>
__device__ __forceinline__ void attn_ro_bwd_entities(const AttnRoRows<M, NU> (&rows)[NE], const Cols<NU> (&go)[NE], const int64_t (&n)[NE], const bool (&live)[NE],
                                                     int D, int H, int g, float inv_sqrt_d, const uint8_t* __restrict__ keep, float kscale,
                                                     float* __restrict__ dq_g, float* __restrict__ dkv, float* __restrict__ ro3, int lane, Cols<NU>& sq,
                                                     Cols<NU>& sk, Cols<NU>& sv) {
    const int d = D / H;
    Cols<NU> dk[NE][M], dv[NE][M], dq[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int u = 0; u < NU; ++u) dk[e][i].v[u] = dv[e][i].v[u] = 0.f;
#pragma unroll
        for (int u = 0; u < NU; ++u) dq[e].v[u] = 0.f;
    }
    for (int h = 0; h < H; ++h) {
        const int lo = h * d, hi = lo + d;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const Cols<NU>& q = rows[e].q;
            const Cols<NU> (&k)[M] = rows[e].k;
            const Cols<NU> (&v)[M] = rows[e].v;
            float p[M], dp[M], ds[M];
            head_scores_row<M, NU>(q, k, lo, hi, lane, inv_sqrt_d, p);
            float rowdot = 0.f;
#pragma unroll
            for (int j = 0; j < M; ++j) {
                float part = 0.f;
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int c = lane + 64 * u;
                    part += (c >= lo && c < hi) ? go[e].v[u] * v[j].v[u] : 0.f;
                }
                const float gg = sum64(part);
                const int kf = entry_at(rows[e].kb, h * M * M + g * M + j);
                const float km = (keep == nullptr) ? 1.f : (kf ? kscale : 0.f);
                const float pt = p[j] * km;
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int c = lane + 64 * u;
                    dv[e][j].v[u] += (c >= lo && c < hi) ? pt * go[e].v[u] : 0.f;
                }
                dp[j] = gg * km;
                rowdot += dp[j] * p[j];
            }
#pragma unroll
            for (int j = 0; j < M; ++j) ds[j] = p[j] * (dp[j] - rowdot) * inv_sqrt_d;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int c = lane + 64 * u;
                const bool in = (c >= lo && c < hi);
                float aq = 0.f;
#pragma unroll
                for (int j = 0; j < M; ++j) {
                    aq += ds[j] * k[j].v[u];
                    dk[e][j].v[u] += in ? ds[j] * q.v[u] : 0.f;
                }
                dq[e].v[u] += in ? aq : 0.f;
            }
        }
    }
    // ro3 != nullptr: the read-out token's dq | dk | dv side by side in ro3 [N, 3D] (one K = 3D operand for d x and d Wqkv) instead of
    // dq_g and the token's own rows of dkv
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        if (!live[e]) continue;
        st_cols<NU>(ro3 ? ro3 + n[e] * (int64_t)(3 * D) : dq_g + n[e] * (int64_t)D, dq[e], D, lane);
#pragma unroll
        for (int u = 0; u < NU; ++u) sq.v[u] += dq[e].v[u];
#pragma unroll
        for (int i = 0; i < M; ++i) {
            float* row = (ro3 && i == g) ? ro3 + n[e] * (int64_t)(3 * D) + D : dkv + (n[e] * M + i) * (int64_t)(2 * D);
            st_cols<NU>(row, dk[e][i], D, lane);
            st_cols<NU>(row + D, dv[e][i], D, lane);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                sk.v[u] += dk[e][i].v[u];
                sv.v[u] += dv[e][i].v[u];
            }
        }
    }
}

template <int M, int NU>
__global__ __launch_bounds__(256) void attn_ro_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dctx_g, int N, int D,
                                                          int H, int g, float inv_sqrt_d, const uint8_t* __restrict__ keep, float kscale,
                                                          float* __restrict__ dq_g, float* __restrict__ dkv, float* __restrict__ bias_part,
                                                          float* __restrict__ ro3) {
    __shared__ float red[4][3][64 * NU];
    const int lane = threadIdx.x & 63;
    const int wave = wave_id_uniform();
    Cols<NU> sq, sk, sv;
#pragma unroll
    for (int u = 0; u < NU; ++u) sq.v[u] = sk.v[u] = sv.v[u] = 0.f;
    for (int64_t n = (int64_t)blockIdx.x * 4 + wave; n < N; n += (int64_t)gridDim.x * 4) {
        const Cols<NU> go = ld_cols<NU>(dctx_g + n * (int64_t)D, D, lane);
        const AttnRoRows<M, NU> rows[1] = {attn_ro_load<M, NU>(qkv, n, D, H, g, keep, lane)};
        const Cols<NU> gos[1] = {go};
        const int64_t ns[1] = {n};
        const bool live[1] = {true};
        attn_ro_bwd_entities<M, NU, 1>(rows, gos, ns, live, D, H, g, inv_sqrt_d, keep, kscale, dq_g, dkv, ro3, lane, sq, sk, sv);
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        red[wave][0][lane + 64 * u] = sq.v[u];
        red[wave][1][lane + 64 * u] = sk.v[u];
        red[wave][2][lane + 64 * u] = sv.v[u];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * 64 * NU; i += 256) {
        const int which = i / (64 * NU), c = i % (64 * NU);
        if (c < D) bias_part[(int64_t)blockIdx.x * 3 * D + which * D + c] = ((red[0][which][c] + red[1][which][c]) + red[2][which][c]) + red[3][which][c];
    }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over the last dim, eps inside the sqrt, biased variance (nn.LayerNorm(eps=1e-12)).
// reference: model/SNAG_tools.py:215,221 and :259,265.   stats[row] = (mean, rstd)
template <int NU>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ t, int64_t R, int D, const float* __restrict__ g,
                                                     const float* __restrict__ b, float eps, float* __restrict__ y,
                                                     float* __restrict__ stats) {
    const int64_t row = (int64_t)blockIdx.x * 4 + wave_id_uniform();
    if (row >= R) return;
    const int lane = threadIdx.x & 63;
    const Cols<NU> x = ld_cols<NU>(t + row * D, D, lane);
    Cols<NU> o;
    float mean, rstd;
    ln_row_forward<NU>(x.v, D, lane, g, b, eps, o.v, mean, rstd);
    st_cols<NU>(y + row * D, o, D, lane);
    if (lane == 0) {
        stats[2 * row] = mean;
        stats[2 * row + 1] = rstd;
    }
}

// dT = rstd (dxhat - mean(dxhat) - xhat mean(dxhat xhat)), dxhat = dY gamma.
// Optional second output dTm = dT * keep * kscale (gradient w.r.t. the pre-dropout dense output).
// Per-block partial sums of dgamma / dbeta / colsum(dTm or dT) (= the bias gradient of the dense in front of
// this LayerNorm) go to part[block][3][D]; a second kernel adds them in order.
template <int NU>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ t,
                                                     const float* __restrict__ stats, const float* __restrict__ g, int64_t R,
                                                     int D, const uint8_t* __restrict__ keep, int64_t keep_ld, float kscale,
                                                     float* __restrict__ dt, float* __restrict__ dtm, float* __restrict__ part,
                                                     float* __restrict__ dy_zero) {
    // keep: row r of the mask starts at keep + r * keep_ld (keep_ld = D dense; M*D when the rows are one token slot of a
    // stacked [N,M,D] mask).  dy_zero (= dy, or NULL): every row of dy is overwritten with zeros once it has been read, so a
    // scatter-add target is clean for the next step without a separate fill launch.
    __shared__ float red[4][3][64 * NU];
    const int lane = threadIdx.x & 63;
    const int wave = wave_id_uniform();
    const Cols<NU> gg = ld_cols<NU>(g, D, lane);
    Cols<NU> ag, ab, az;
#pragma unroll
    for (int u = 0; u < NU; ++u) ag.v[u] = ab.v[u] = az.v[u] = 0.f;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < R; row += (int64_t)gridDim.x * 4) {
        const Cols<NU> x = ld_cols<NU>(t + row * D, D, lane);
        const Cols<NU> gy = ld_cols<NU>(dy + row * D, D, lane);
        if (dy_zero != nullptr) {
            Cols<NU> z;
#pragma unroll
            for (int u = 0; u < NU; ++u) z.v[u] = 0.f;
            st_cols<NU>(dy_zero + row * D, z, D, lane);
        }
        const float mean = stats[2 * row], rstd = stats[2 * row + 1];
        Cols<NU> xh, dxh;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const bool in = lane + 64 * u < D;
            xh.v[u] = in ? (x.v[u] - mean) * rstd : 0.f;
            dxh.v[u] = gy.v[u] * gg.v[u];
            s1 += dxh.v[u];
            s2 = fmaf(dxh.v[u], xh.v[u], s2);
            ag.v[u] = fmaf(gy.v[u], xh.v[u], ag.v[u]);
            ab.v[u] += gy.v[u];
        }
        s1 = sum64(s1) / (float)D;
        s2 = sum64(s2) / (float)D;
        Cols<NU> o;
#pragma unroll
        for (int u = 0; u < NU; ++u) o.v[u] = rstd * (dxh.v[u] - s1 - xh.v[u] * s2);
        st_cols<NU>(dt + row * D, o, D, lane);
        if (dtm != nullptr) {
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int c = lane + 64 * u;
                const float z = (c < D && keep[row * keep_ld + c]) ? o.v[u] * kscale : 0.f;
                if (c < D) dtm[row * D + c] = z;
                az.v[u] += z;
            }
        } else {
#pragma unroll
            for (int u = 0; u < NU; ++u) az.v[u] += o.v[u];
        }
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        red[wave][0][lane + 64 * u] = ag.v[u];
        red[wave][1][lane + 64 * u] = ab.v[u];
        red[wave][2][lane + 64 * u] = az.v[u];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * 64 * NU; i += 256) {
        const int which = i / (64 * NU), c = i % (64 * NU);
        if (c < D) part[((int64_t)blockIdx.x * 3 + which) * D + c] = ((red[0][which][c] + red[1][which][c]) + red[2][which][c]) + red[3][which][c];
    }
}

// out[c] (=|+=) sum over blocks of part[block][c].  16 waves each add a contiguous run of partials for
// 64 columns, then the 16 sub-sums are combined in wave order: deterministic, and 16x shorter than one
// serial chain per column.
static __global__ __launch_bounds__(1024) void slab_colsum_kernel(const float* __restrict__ part, int nblocks, int64_t stride, int ncol,
                                                                   float* __restrict__ out, int accumulate) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, wave = wave_id_uniform();
    const int c = blockIdx.x * 64 + lane;
    const int per = (nblocks + 15) / 16;
    const int b0 = wave * per, b1 = min(nblocks, b0 + per);
    float s = 0.f;
    if (c < ncol) {
        int b = b0;
        for (; b + 8 <= b1; b += 8) {                  // eight loads in flight, added in order
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = part[(int64_t)(b + u) * stride + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += t[u];
        }
        for (; b < b1; ++b) s += part[(int64_t)b * stride + c];
    }
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && c < ncol) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += red[w][lane];
        out[c] = accumulate ? out[c] + t : t;
    }
}

// The same reduction for LayerNorm's [blocks][3][D] partials, written straight to their three destinations (dgamma, dbeta and the
// bias gradient of the dense layer in front of the LayerNorm): column c of the 3D-wide slab goes to a[c], b[c - D] or c_[c - 2D].
static __global__ __launch_bounds__(1024) void slab_colsum3_kernel(const float* __restrict__ part, int nblocks, int D, float* __restrict__ a,
                                                                    float* __restrict__ b, float* __restrict__ c_, int accumulate) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, wave = wave_id_uniform();
    const int c = blockIdx.x * 64 + lane;
    const int ncol = 3 * D;
    const int64_t stride = ncol;
    const int per = (nblocks + 15) / 16;
    const int b0 = wave * per, b1 = min(nblocks, b0 + per);
    float s = 0.f;
    if (c < ncol) {
        int blk = b0;
        for (; blk + 8 <= b1; blk += 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = part[(int64_t)(blk + u) * stride + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += t[u];
        }
        for (; blk < b1; ++blk) s += part[(int64_t)blk * stride + c];
    }
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && c < ncol) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += red[w][lane];
        float* dst = c < D ? a + c : (c < 2 * D ? b + (c - D) : c_ + (c - 2 * D));
        *dst = accumulate ? *dst + t : t;
    }
}

// column sums of a [R, C] matrix (bias gradients): part[blockIdx.y][c]
static __global__ __launch_bounds__(256) void colsum_part_kernel(const float* __restrict__ x, int64_t R, int C, int64_t ld,
                                                          float* __restrict__ part) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = wave_id_uniform();
    const int c = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (c < C)
        for (int64_t r = (int64_t)blockIdx.y * 4 + wave; r < R; r += (int64_t)gridDim.y * 4) s += x[r * ld + c];
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && c < C) part[(int64_t)blockIdx.y * C + c] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
}

// ---------------------------------------------------------------------------------------------
// Fusion epilogue.  reference: MformerFusion.forward, model/SNAG_tools.py:41-49
//   c_j = sum_h sum_i P~[h,i,j] / sqrt(M H) ; w = softmax(c) ; joint_m = w_m nrm(x_m) ;
//   joint_fz_m = softmax(weight_raw[0:6])_m nrm(x_m)        (x_m = the ORIGINAL token m)
template <int M, int NU>
__global__ __launch_bounds__(256) void fuse_out_fwd_kernel(const float* __restrict__ probs, const float* __restrict__ x0,
                                                           const float* __restrict__ weight_raw, int N, int D, int H,
                                                           float* __restrict__ joint, float* __restrict__ joint_fz,
                                                           float* __restrict__ wnorm) {
    const int64_t n = (int64_t)blockIdx.x * 4 + wave_id_uniform();
    if (n >= N) return;
    const int lane = threadIdx.x & 63;
    float w[M];
    {
        const float inv = 1.0f / sqrtf((float)(M * H));
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < M; ++j) {
            float s = 0.f;
            for (int h = 0; h < H; ++h)
#pragma unroll
                for (int i = 0; i < M; ++i) s += probs[((n * H + h) * M + i) * (int64_t)M + j];
            w[j] = s * inv;
            mx = fmaxf(mx, w[j]);
        }
        float den = 0.f;
#pragma unroll
        for (int j = 0; j < M; ++j) {
            w[j] = expf(w[j] - mx);
            den += w[j];
        }
#pragma unroll
        for (int j = 0; j < M; ++j) w[j] /= den;
    }
    float wf[6];
    {
        float mx = -INFINITY, den = 0.f;
#pragma unroll
        for (int j = 0; j < 6; ++j) mx = fmaxf(mx, weight_raw[j]);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            wf[j] = expf(weight_raw[j] - mx);
            den += wf[j];
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) wf[j] /= den;
    }
    if (wnorm != nullptr && lane < M) {
        float mine = 0.f;
#pragma unroll
        for (int j = 0; j < M; ++j) mine = (lane == j) ? w[j] : mine;
        wnorm[n * M + lane] = mine;
    }
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const Cols<NU> x = ld_cols<NU>(x0 + (n * M + m) * (int64_t)D, D, lane);
        float ss = 0.f;
#pragma unroll
        for (int u = 0; u < NU; ++u) ss = fmaf(x.v[u], x.v[u], ss);
        const float inv = 1.0f / l2_denominator(sum64(ss));
        Cols<NU> a, b;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const float xh = x.v[u] * inv;
            a.v[u] = w[m] * xh;
            b.v[u] = wf[m] * xh;
        }
        if (joint != nullptr) st_cols<NU>(joint + (n * M + m) * (int64_t)D, a, D, lane);
        if (joint_fz != nullptr) st_cols<NU>(joint_fz + (n * M + m) * (int64_t)D, b, D, lane);
    }
}

// backward: (d_joint, d_joint_fz, d_wnorm) -> d_x0 [N,M,D], dprobs_extra [N,H,M,M], partial d(softmax(weight_raw))
// part[block][6] holds sum_n d_joint_fz_m . nrm(x_m) for the block's entities (zero for m >= M).
template <int M, int NU>
__global__ __launch_bounds__(256) void fuse_out_bwd_kernel(const float* __restrict__ probs, const float* __restrict__ x0,
                                                           const float* __restrict__ weight_raw, int N, int D, int H,
                                                           const float* __restrict__ d_joint, const float* __restrict__ d_joint_fz,
                                                           const float* __restrict__ d_wnorm, float* __restrict__ d_x0,
                                                           float* __restrict__ dprobs_extra, float* __restrict__ part) {
    __shared__ float red[4][6];
    const int lane = threadIdx.x & 63, wave = wave_id_uniform();
    float wf[6];
    {
        float mx = -INFINITY, den = 0.f;
#pragma unroll
        for (int j = 0; j < 6; ++j) mx = fmaxf(mx, weight_raw[j]);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            wf[j] = expf(weight_raw[j] - mx);
            den += wf[j];
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) wf[j] /= den;
    }
    float dwf[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int64_t n = (int64_t)blockIdx.x * 4 + wave; n < N; n += (int64_t)gridDim.x * 4) {
        float w[M], dw[M];
        {
            const float inv = 1.0f / sqrtf((float)(M * H));
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < M; ++j) {
                float s = 0.f;
                for (int h = 0; h < H; ++h)
#pragma unroll
                    for (int i = 0; i < M; ++i) s += probs[((n * H + h) * M + i) * (int64_t)M + j];
                w[j] = s * inv;
                mx = fmaxf(mx, w[j]);
            }
            float den = 0.f;
#pragma unroll
            for (int j = 0; j < M; ++j) {
                w[j] = expf(w[j] - mx);
                den += w[j];
            }
#pragma unroll
            for (int j = 0; j < M; ++j) w[j] /= den;
        }
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const int64_t off = (n * M + m) * (int64_t)D;
            Cols<NU> x = ld_cols<NU>(x0 + off, D, lane);
            float ss = 0.f;
#pragma unroll
            for (int u = 0; u < NU; ++u) ss = fmaf(x.v[u], x.v[u], ss);
            const float den = l2_denominator(sum64(ss));
            const float inv = 1.0f / den;
            Cols<NU> gj, gf, dxh;
#pragma unroll
            for (int u = 0; u < NU; ++u) gj.v[u] = gf.v[u] = 0.f;
            if (d_joint != nullptr) gj = ld_cols<NU>(d_joint + off, D, lane);
            if (d_joint_fz != nullptr) gf = ld_cols<NU>(d_joint_fz + off, D, lane);
            float a = 0.f, b = 0.f, dot = 0.f;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                x.v[u] *= inv;                                   // xhat
                a = fmaf(gj.v[u], x.v[u], a);                    // d w_m
                b = fmaf(gf.v[u], x.v[u], b);                    // d wfz_m
                dxh.v[u] = w[m] * gj.v[u] + wf[m] * gf.v[u];     // d xhat
                dot = fmaf(dxh.v[u], x.v[u], dot);
            }
            a = sum64(a);
            b = sum64(b);
            dot = (den > 1e-12f) ? sum64(dot) : 0.f;
            dw[m] = a + (d_wnorm != nullptr ? d_wnorm[n * M + m] : 0.f);
            dwf[m] += b;
            Cols<NU> o;
#pragma unroll
            for (int u = 0; u < NU; ++u) o.v[u] = (dxh.v[u] - x.v[u] * dot) * inv;
            st_cols<NU>(d_x0 + off, o, D, lane);
        }
        // softmax backward to c, then c_j -> every P~[h,i,j]
        float sdot = 0.f;
#pragma unroll
        for (int j = 0; j < M; ++j) sdot += w[j] * dw[j];
        const float inv = 1.0f / sqrtf((float)(M * H));
        for (int e = lane; e < H * M * M; e += 64) {   // element e = (h, i, j), j = e % M
            const int j = e % M;
            float val = 0.f;
#pragma unroll
            for (int jj = 0; jj < M; ++jj) val = (j == jj) ? w[jj] * (dw[jj] - sdot) * inv : val;
            dprobs_extra[n * (int64_t)(H * M * M) + e] = val;
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 6; ++j) red[wave][j] = dwf[j];
    }
    __syncthreads();
    if (threadIdx.x < 6) part[(int64_t)blockIdx.x * 6 + threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// d weight_raw = softmax backward over the 6 raw weights (reference softmaxes all 6, SNAG_tools.py:46)
static __global__ __launch_bounds__(64) void weight_raw_bwd_kernel(const float* __restrict__ part, int nblocks, const float* __restrict__ weight_raw,
                                                                   float* __restrict__ d_weight_raw, int accumulate) {
    // one wave: lane l sums blocks l, l+64, ... in order, then a fixed-pattern wave reduction (deterministic)
    // (the six columns of a block row are requested together and four rows are in flight per lane: as six separate column loops the launch was
    // 96 dependent round trips -- 12.6 us; the order of every column's sum is unchanged)
    float g[6], s6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int b = threadIdx.x; b < nblocks; b += 64) {
        float v[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) v[j] = part[(int64_t)b * 6 + j];
#pragma unroll
        for (int j = 0; j < 6; ++j) s6[j] += v[j];
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) g[j] = sum64(s6[j]);
    if (threadIdx.x != 0) return;
    float wf[6];
    float mx = -INFINITY, den = 0.f;
    for (int j = 0; j < 6; ++j) mx = fmaxf(mx, weight_raw[j]);
    for (int j = 0; j < 6; ++j) {
        wf[j] = expf(weight_raw[j] - mx);
        den += wf[j];
    }
    float sdot = 0.f;
    for (int j = 0; j < 6; ++j) {
        wf[j] /= den;
        sdot += wf[j] * g[j];
    }
    for (int j = 0; j < 6; ++j) {
        const float v = wf[j] * (g[j] - sdot);
        d_weight_raw[j] = accumulate ? d_weight_raw[j] + v : v;
    }
}

// ---------------------------------------------------------------------------------------------
// Token packing: out[n, m, :] = emb_m[n, :]  (torch.stack(embs, dim=1), model/SNAG_tools.py:36), with the
// entity-noise blend folded into the structural slot g (model/SNAG_tools.py:127-128):
//   e = (1 - a) e + a noise   for rows with mask, a = mask_ratio / 2.
struct TokenPtrs {
    const float* p[6];
};

static __global__ __launch_bounds__(256) void pack_tokens_kernel(TokenPtrs src, int M, int64_t N, int D, int g_slot,
                                                          const float* __restrict__ noise, const uint8_t* __restrict__ nmask,
                                                          float a, float* __restrict__ out) {
    const int64_t total = N * M * (int64_t)D;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % D);
        const int64_t nm = i / D;
        const int m = (int)(nm % M);
        const int64_t n = nm / M;
        float v = src.p[m][n * D + c];
        if (m == g_slot && nmask != nullptr && nmask[n]) v = (1.0f - a) * v + a * noise[n * D + c];
        out[i] = v;
    }
}

// hd_mean read-out and its backward (README joint_way "Mformer_hd_mean": hidden.mean(dim=1))
static __global__ __launch_bounds__(256) void token_mean_kernel(const float* __restrict__ hidden, int64_t N, int M, int D,
                                                         float* __restrict__ out) {
    const int64_t total = N * (int64_t)D;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t n = i / D;
        const int c = (int)(i % D);
        float s = 0.f;
        for (int m = 0; m < M; ++m) s += hidden[(n * M + m) * D + c];
        out[i] = s / (float)M;
    }
}

}  // namespace snag
