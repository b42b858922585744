// SURVEY.md par. 8(f) "next" #2: the sparse GAT structural encoder, diag variant, as SNAG instantiates it
// (MultiHeadGraphAttention model/layers.py:35-100; GAT model/Tool_model.py:61-94; eval / zero-dropout configuration).
//
// One layer, H heads (averaged), input x [N, f]:
//   h_i = x * w_i                       (w_i [f], the diag weight)
//   e_i(u,v) = exp(-leaky_relu_0.2(a_src_i . h_i[u] + a_dst_i . h_i[v]))          per stored edge (u, v)
//   A_i[u,:] = sum_{(u,v)} e_i(u,v) x[v,:] ,  r_i[u] = sum_{(u,v)} e_i(u,v)
//   pre[u,:] = (1/H) sum_i w_i * A_i[u,:] / r_i[u] ;  out = elu(pre) between layers, pre after the last.
//
// Training options of the reference (round 5): edge dropout -- F.dropout(edge_e, attn_dropout) AFTER the row sums (layers.py:81-83): the aggregation
// uses e * keep / (1 - p), the normaliser r the undropped e -- as an optional byte mask keep[i, stored edge] read through the CSR position ->
// stored-edge-id arrays; the input dropout of the non-final layers (Tool_model.py:78-79) is an elementwise launch of its own (snag_keep_scale).
//
// Edges arrive as CSR over the source node (duplicates and self loops preserved, values unused).  One wave per source
// node; a feature row is spread over the wave as col = lane + 64*u (256-byte segments).  The backward is two gather passes (out-edges per
// source, in-edges per destination) with no atomics; parameter gradients go through fixed-order per-block partial sums.
#include <algorithm>

#include "fusion_kernels.hip.h"

namespace snag {

constexpr int kGatMaxH = 4;

// s_src[n,i] = sum_c x[n,c] w_i[c] a_i[c] ; s_dst[n,i] = sum_c x[n,c] w_i[c] a_i[f + c]
template <int NU>
__global__ __launch_bounds__(256) void gat_node_scores_kernel(const float* __restrict__ x, int64_t N, int f, int H, const float* __restrict__ w,
                                                              const float* __restrict__ a, float* __restrict__ s_src, float* __restrict__ s_dst) {
    const int64_t n = (int64_t)blockIdx.x * 4 + wave_id_uniform();
    if (n >= N) return;
    const int lane = threadIdx.x & 63;
    const Cols<NU> xr = ld_cols<NU>(x + n * f, f, lane);
    for (int i = 0; i < H; ++i) {
        const Cols<NU> wi = ld_cols<NU>(w + i * f, f, lane);
        const Cols<NU> as = ld_cols<NU>(a + i * 2 * f, f, lane), ad = ld_cols<NU>(a + i * 2 * f + f, f, lane);
        float ps = 0.f, pd = 0.f;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const float h = xr.v[u] * wi.v[u];
            ps = fmaf(h, as.v[u], ps);
            pd = fmaf(h, ad.v[u], pd);
        }
        ps = sum64(ps);
        pd = sum64(pd);
        if (lane == 0) {
            s_src[n * H + i] = ps;
            s_dst[n * H + i] = pd;
        }
    }
}

__device__ __forceinline__ float gat_edge_weight(float z) { return expf(-(z > 0.f ? z : 0.2f * z)); }

// forward aggregation; agg [N,H,f] and rsum [N,H] are kept for the backward
template <int NU>
__global__ __launch_bounds__(256) void gat_aggregate_fwd_kernel(const float* __restrict__ x, int64_t N, int f, int H, const float* __restrict__ w,
                                                                const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                                const float* __restrict__ s_src, const float* __restrict__ s_dst, int apply_elu,
                                                                float* __restrict__ agg, float* __restrict__ rsum, float* __restrict__ pre,
                                                                float* __restrict__ out, const uint8_t* __restrict__ edge_keep,
                                                                const int32_t* __restrict__ eid, int64_t n_edges, float escale) {
    const int64_t u = (int64_t)blockIdx.x * 4 + wave_id_uniform();
    if (u >= N) return;
    const int lane = threadIdx.x & 63;
    Cols<NU> acc[kGatMaxH];
    float rs[kGatMaxH], ss[kGatMaxH];
#pragma unroll
    for (int i = 0; i < kGatMaxH; ++i) {
        rs[i] = 0.f;
        ss[i] = i < H ? s_src[u * H + i] : 0.f;
#pragma unroll
        for (int q = 0; q < NU; ++q) acc[i].v[q] = 0.f;
    }
    const int e0 = rowptr[u], e1 = rowptr[u + 1];
    for (int e = e0; e < e1; ++e) {
        const int64_t v = col[e];
        const Cols<NU> xv = ld_cols<NU>(x + v * f, f, lane);
#pragma unroll
        for (int i = 0; i < kGatMaxH; ++i) {
            if (i < H) {
                const float al = gat_edge_weight(ss[i] + s_dst[v * H + i]);
                rs[i] += al;                                                           // the row sum takes the undropped weight (:81)
                const float ald = edge_keep == nullptr ? al : (edge_keep[(int64_t)i * n_edges + eid[e]] ? al * escale : 0.f);      // F.dropout(edge_e) (:83)
#pragma unroll
                for (int q = 0; q < NU; ++q) acc[i].v[q] = fmaf(ald, xv.v[q], acc[i].v[q]);
            }
        }
    }
    Cols<NU> o;
#pragma unroll
    for (int q = 0; q < NU; ++q) o.v[q] = 0.f;
#pragma unroll
    for (int i = 0; i < kGatMaxH; ++i) {
        if (i < H) {
            st_cols<NU>(agg + (u * H + i) * f, acc[i], f, lane);
            if (lane == 0) rsum[u * H + i] = rs[i];
            const Cols<NU> wi = ld_cols<NU>(w + i * f, f, lane);
            const float inv = 1.0f / rs[i];
#pragma unroll
            for (int q = 0; q < NU; ++q) o.v[q] += wi.v[q] * acc[i].v[q] * inv;       // h'_i = (w_i * A_i) / r_i   (:85-86)
        }
    }
    const float invH = 1.0f / (float)H;
    Cols<NU> y;
#pragma unroll
    for (int q = 0; q < NU; ++q) {
        o.v[q] *= invH;                                                                // mean over heads (Tool_model.py:84-85)
        y.v[q] = (apply_elu && o.v[q] <= 0.f) ? expm1f(o.v[q]) : o.v[q];               // F.elu (:87-88)
    }
    if (pre != nullptr) st_cols<NU>(pre + u * f, o, f, lane);
    st_cols<NU>(out + u * f, y, f, lane);
}

// Backward, pass 1: one wave per source node u over its OUT-edges (persistent blocks, fixed-order partials).
//   gq[u,:]   = d_out[u,:] * elu'(pre[u,:])                      (kept for pass 2)
//   drs[u,i]  = d loss / d r_i[u] = -sum_c gq w_i A_i / (H r^2)   (kept for pass 2)
//   ds_src[u,i] = sum_{(u,v)} dz_i(u,v),  dz = -e * lrelu'(z) * (dA_i[u,:] . x[v,:] + drs),  dA_i[u,:] = gq w_i / (H r_i[u])
//   part[block][H][f]: the block's partial of sum_u gq[u,:] A_i[u,:] / (H r_i[u])   (the aggregation route into d w_i)
template <int NU>
__global__ __launch_bounds__(256) void gat_bwd_out_edges_kernel(const float* __restrict__ x, int64_t N, int f, int H, const float* __restrict__ w,
                                                                const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                                const float* __restrict__ s_src, const float* __restrict__ s_dst,
                                                                const float* __restrict__ agg, const float* __restrict__ rsum,
                                                                const float* __restrict__ pre, int had_elu, const float* __restrict__ d_out,
                                                                float* __restrict__ gq, float* __restrict__ drs_out, float* __restrict__ ds_src,
                                                                float* __restrict__ part, const uint8_t* __restrict__ edge_keep,
                                                                const int32_t* __restrict__ eid, int64_t n_edges, float escale) {
    __shared__ float red[4][64 * NU];
    const int lane = threadIdx.x & 63, wave = wave_id_uniform();
    Cols<NU> pw[kGatMaxH], wi[kGatMaxH];
#pragma unroll
    for (int i = 0; i < kGatMaxH; ++i) {
        if (i < H) wi[i] = ld_cols<NU>(w + i * f, f, lane);
#pragma unroll
        for (int q = 0; q < NU; ++q) pw[i].v[q] = 0.f;
    }
    const float invH = 1.0f / (float)H;
    for (int64_t u = (int64_t)blockIdx.x * 4 + wave; u < N; u += (int64_t)gridDim.x * 4) {
        Cols<NU> g = ld_cols<NU>(d_out + u * f, f, lane);
        if (had_elu) {
            const Cols<NU> p = ld_cols<NU>(pre + u * f, f, lane);
#pragma unroll
            for (int q = 0; q < NU; ++q) g.v[q] *= (p.v[q] > 0.f) ? 1.0f : expf(p.v[q]);      // elu'
        }
        st_cols<NU>(gq + u * f, g, f, lane);
        Cols<NU> dA[kGatMaxH];                        // d loss / d A_i[u,:] = g w_i / (H r)
        float drs[kGatMaxH], ssu[kGatMaxH], dsu[kGatMaxH];
#pragma unroll
        for (int i = 0; i < kGatMaxH; ++i) {
            drs[i] = ssu[i] = dsu[i] = 0.f;
            if (i < H) {
                const Cols<NU> A = ld_cols<NU>(agg + (u * H + i) * f, f, lane);
                const float inv = 1.0f / rsum[u * H + i];
                float t = 0.f;
#pragma unroll
                for (int q = 0; q < NU; ++q) {
                    const float gw = g.v[q] * invH * wi[i].v[q] * inv;
                    dA[i].v[q] = gw;
                    t = fmaf(-gw * inv, A.v[q], t);
                    pw[i].v[q] = fmaf(g.v[q] * invH * inv, A.v[q], pw[i].v[q]);
                }
                drs[i] = sum64(t);
                ssu[i] = s_src[u * H + i];
                if (lane == 0) drs_out[u * H + i] = drs[i];
            }
        }
        const int e0 = rowptr[u], e1 = rowptr[u + 1];
        for (int e = e0; e < e1; ++e) {
            const int64_t v = col[e];
            const Cols<NU> xv = ld_cols<NU>(x + v * f, f, lane);
#pragma unroll
            for (int i = 0; i < kGatMaxH; ++i) {
                if (i < H) {
                    const float z = ssu[i] + s_dst[v * H + i];
                    float dal = 0.f;
#pragma unroll
                    for (int q = 0; q < NU; ++q) dal = fmaf(dA[i].v[q], xv.v[q], dal);
                    const float ks = edge_keep == nullptr ? 1.f : (edge_keep[(int64_t)i * n_edges + eid[e]] ? escale : 0.f);
                    dal = ks * sum64(dal) + drs[i];                           // through A_i (the dropped weight) and through r_i (the undropped one)
                    dsu[i] += -gat_edge_weight(z) * (z > 0.f ? 1.0f : 0.2f) * dal;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < kGatMaxH; ++i)
            if (i < H && lane == 0) ds_src[u * H + i] = dsu[i];
    }
    for (int i = 0; i < H; ++i) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NU; ++q) {
            float val = 0.f;
#pragma unroll
            for (int j = 0; j < kGatMaxH; ++j) val = (j == i) ? pw[j].v[q] : val;
            red[wave][lane + 64 * q] = val;
        }
        __syncthreads();
        for (int c = threadIdx.x; c < f; c += 256) part[((int64_t)blockIdx.x * H + i) * f + c] = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
    }
}

// Backward, pass 2: one wave per destination node v over its IN-edges (t_rowptr/t_col: the same edge set, grouped by destination).
// No atomics: d_x[v,:] = sum_i w_i * sum_{(u,v)} e_i(u,v) gq[u,:] / (H r_i[u]) and ds_dst[v,i] = sum_{(u,v)} dz_i(u,v) are owned by this wave.
template <int NU>
__global__ __launch_bounds__(256) void gat_bwd_in_edges_kernel(const float* __restrict__ x, int64_t N, int f, int H, const float* __restrict__ w,
                                                               const int32_t* __restrict__ t_rowptr, const int32_t* __restrict__ t_col,
                                                               const float* __restrict__ s_src, const float* __restrict__ s_dst,
                                                               const float* __restrict__ rsum, const float* __restrict__ gq,
                                                               const float* __restrict__ drs, float* __restrict__ d_x, float* __restrict__ ds_dst,
                                                               const uint8_t* __restrict__ edge_keep, const int32_t* __restrict__ t_eid, int64_t n_edges,
                                                               float escale) {
    const int64_t v = (int64_t)blockIdx.x * 4 + wave_id_uniform();
    if (v >= N) return;
    const int lane = threadIdx.x & 63;
    const float invH = 1.0f / (float)H;
    const Cols<NU> xv = ld_cols<NU>(x + v * f, f, lane);
    Cols<NU> xw[kGatMaxH], acc[kGatMaxH];
    float sd[kGatMaxH], dsd[kGatMaxH];
#pragma unroll
    for (int i = 0; i < kGatMaxH; ++i) {
        sd[i] = dsd[i] = 0.f;
        if (i < H) {
            const Cols<NU> wi = ld_cols<NU>(w + i * f, f, lane);
            sd[i] = s_dst[v * H + i];
#pragma unroll
            for (int q = 0; q < NU; ++q) xw[i].v[q] = xv.v[q] * wi.v[q];
        }
#pragma unroll
        for (int q = 0; q < NU; ++q) acc[i].v[q] = 0.f;
    }
    const int e0 = t_rowptr[v], e1 = t_rowptr[v + 1];
    for (int e = e0; e < e1; ++e) {
        const int64_t u = t_col[e];
        const Cols<NU> g = ld_cols<NU>(gq + u * f, f, lane);
#pragma unroll
        for (int i = 0; i < kGatMaxH; ++i) {
            if (i < H) {
                const float z = s_src[u * H + i] + sd[i];
                const float al = gat_edge_weight(z), c = invH / rsum[u * H + i];
                const float ks = edge_keep == nullptr ? 1.f : (edge_keep[(int64_t)i * n_edges + t_eid[e]] ? escale : 0.f);
                float dot = 0.f;
#pragma unroll
                for (int q = 0; q < NU; ++q) {
                    acc[i].v[q] = fmaf(al * ks * c, g.v[q], acc[i].v[q]);
                    dot = fmaf(g.v[q], xw[i].v[q], dot);
                }
                const float dal = ks * c * sum64(dot) + drs[u * H + i];
                dsd[i] += -al * (z > 0.f ? 1.0f : 0.2f) * dal;
            }
        }
    }
    Cols<NU> o;
#pragma unroll
    for (int q = 0; q < NU; ++q) o.v[q] = 0.f;
#pragma unroll
    for (int i = 0; i < kGatMaxH; ++i) {
        if (i < H) {
            const Cols<NU> wi = ld_cols<NU>(w + i * f, f, lane);
#pragma unroll
            for (int q = 0; q < NU; ++q) o.v[q] = fmaf(wi.v[q], acc[i].v[q], o.v[q]);
            if (lane == 0) ds_dst[v * H + i] = dsd[i];
        }
    }
    st_cols<NU>(d_x + v * f, o, f, lane);
}

// Node pass of the backward: s_src/s_dst -> x, w, a.  part2[block][H][3][f]: partials of d w_i (score route), d a_src_i, d a_dst_i.
template <int NU>
__global__ __launch_bounds__(256) void gat_node_scores_bwd_kernel(const float* __restrict__ x, int64_t N, int f, int H, const float* __restrict__ w,
                                                                  const float* __restrict__ a, const float* __restrict__ ds_src,
                                                                  const float* __restrict__ ds_dst, float* __restrict__ d_x,
                                                                  float* __restrict__ part2) {
    __shared__ float red[4][3][64 * NU];
    const int lane = threadIdx.x & 63, wave = wave_id_uniform();
    for (int i = 0; i < H; ++i) {
        const Cols<NU> wi = ld_cols<NU>(w + i * f, f, lane);
        const Cols<NU> as = ld_cols<NU>(a + i * 2 * f, f, lane), ad = ld_cols<NU>(a + i * 2 * f + f, f, lane);
        Cols<NU> pw, pas, pad;
#pragma unroll
        for (int q = 0; q < NU; ++q) pw.v[q] = pas.v[q] = pad.v[q] = 0.f;
        for (int64_t n = (int64_t)blockIdx.x * 4 + wave; n < N; n += (int64_t)gridDim.x * 4) {
            const Cols<NU> xr = ld_cols<NU>(x + n * f, f, lane);
            const float gs = ds_src[n * H + i], gd = ds_dst[n * H + i];
#pragma unroll
            for (int q = 0; q < NU; ++q) {
                const int c = lane + 64 * q;
                const float t = gs * as.v[q] + gd * ad.v[q];
                if (c < f) d_x[n * f + c] += t * wi.v[q];                 // this wave owns row n in this kernel: plain read-modify-write
                pw.v[q] = fmaf(t, xr.v[q], pw.v[q]);
                pas.v[q] = fmaf(gs * xr.v[q], wi.v[q], pas.v[q]);
                pad.v[q] = fmaf(gd * xr.v[q], wi.v[q], pad.v[q]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NU; ++q) {
            red[wave][0][lane + 64 * q] = pw.v[q];
            red[wave][1][lane + 64 * q] = pas.v[q];
            red[wave][2][lane + 64 * q] = pad.v[q];
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < 3 * f; idx += 256) {
            const int k = idx / f, c = idx % f;
            part2[(((int64_t)blockIdx.x * H + i) * 3 + k) * f + c] = ((red[0][k][c] + red[1][k][c]) + red[2][k][c]) + red[3][k][c];
        }
    }
}

// d w[i,c] = sum_blocks part[b][i][c] + part2[b][i][0][c] ; d a[i, c] = sum part2[b][i][1][c] ; d a[i, f + c] = sum part2[b][i][2][c]
__global__ __launch_bounds__(256) void gat_param_reduce_kernel(const float* __restrict__ part, const float* __restrict__ part2, int nb1, int nb2, int H,
                                                               int f, float* __restrict__ d_w, float* __restrict__ d_a, int accumulate) {
    // 16 columns x 16 groups of partial blocks per workgroup; group g sums blocks g, g+16, ... in order, then the groups are
    // folded in a fixed order through LDS.
    __shared__ float red[3][16][16];
    const int cl = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int idx = blockIdx.x * 16 + cl;
    float sw = 0.f, sa = 0.f, sd = 0.f;
    if (idx < H * f) {
        const int i = idx / f, c = idx % f;
        for (int b = g; b < nb1; b += 16) sw += part[((int64_t)b * H + i) * f + c];
        for (int b = g; b < nb2; b += 16) {
            const float* q = part2 + (((int64_t)b * H + i) * 3) * f + c;
            sw += q[0];
            sa += q[f];
            sd += q[2 * f];
        }
    }
    red[0][g][cl] = sw;
    red[1][g][cl] = sa;
    red[2][g][cl] = sd;
    __syncthreads();
    if (g != 0 || idx >= H * f) return;
    sw = sa = sd = 0.f;
    for (int k = 0; k < 16; ++k) {
        sw += red[0][k][cl];
        sa += red[1][k][cl];
        sd += red[2][k][cl];
    }
    const int i = idx / f, c = idx % f;
    d_w[idx] = accumulate ? d_w[idx] + sw : sw;
    float* pa = d_a + (int64_t)i * 2 * f + c;
    pa[0] = accumulate ? pa[0] + sa : sa;
    pa[f] = accumulate ? pa[f] + sd : sd;
}

// out[u,:] = sum_{e in row u} val[e] * x[col[e],:]  (+ bias): the torch.spmm(adj, support) of GraphConvolution.forward
// (model/layers.py:124-130).  One wave per row, a feature row spread over the wave; duplicates in the edge list simply add up.
template <int NU>
__global__ __launch_bounds__(256) void spmm_csr_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                       const float* __restrict__ val, const float* __restrict__ x, int64_t N, int f,
                                                       const float* __restrict__ bias, float* __restrict__ out) {
    const int64_t u = (int64_t)blockIdx.x * 4 + wave_id_uniform();
    if (u >= N) return;
    const int lane = threadIdx.x & 63;
    Cols<NU> acc;
#pragma unroll
    for (int q = 0; q < NU; ++q) acc.v[q] = 0.f;
    const int e0 = rowptr[u], e1 = rowptr[u + 1];
    for (int e = e0; e < e1; ++e) {
        const float a = val[e];
        const Cols<NU> xv = ld_cols<NU>(x + (int64_t)col[e] * f, f, lane);
#pragma unroll
        for (int q = 0; q < NU; ++q) acc.v[q] = fmaf(a, xv.v[q], acc.v[q]);
    }
    if (bias) {
        const Cols<NU> b = ld_cols<NU>(bias, f, lane);
#pragma unroll
        for (int q = 0; q < NU; ++q) acc.v[q] += b.v[q];
    }
    st_cols<NU>(out + u * f, acc, f, lane);
}

// F.dropout with an explicit keep mask: out = keep ? x * scale : 0  (forward: x -> the dropped layer input; backward: d of it -> d x)
__global__ __launch_bounds__(256) void keep_scale_kernel(const float* __restrict__ x, int64_t n, const uint8_t* __restrict__ keep, float scale,
                                                         float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = keep[i] ? x[i] * scale : 0.f;
}

static constexpr int kGatBlocks = 1024;

}  // namespace snag

using namespace snag;

#define SNAG_GAT_NU(KERNEL, f, grid, st, ...)                                                        \
    do {                                                                                             \
        if ((f) <= 128) hipLaunchKernelGGL((KERNEL<2>), grid, dim3(256), 0, st, __VA_ARGS__);        \
        else if ((f) <= 320) hipLaunchKernelGGL((KERNEL<5>), grid, dim3(256), 0, st, __VA_ARGS__);   \
        else hipLaunchKernelGGL((KERNEL<8>), grid, dim3(256), 0, st, __VA_ARGS__);                   \
    } while (0)

static int gat_check(int64_t N, int f, int H, const void* x, const void* w, const void* a, const void* rowptr, const void* col) {
    SNAG_REQUIRE(N >= 0 && f >= 1 && H >= 1, SNAG_ERR_ARG, "gat: bad sizes N=%ld f=%d H=%d", (long)N, f, H);
    SNAG_REQUIRE(H <= kGatMaxH, SNAG_ERR_UNSUPPORTED, "gat: %d heads > %d", H, kGatMaxH);
    SNAG_REQUIRE(f <= 512, SNAG_ERR_UNSUPPORTED, "gat: feature width %d > 512", f);
    SNAG_REQUIRE(N < (1LL << 31), SNAG_ERR_UNSUPPORTED, "gat: N >= 2^31");
    if (N == 0) return SNAG_OK;
    SNAG_REQUIRE(x && w && a && rowptr && col, SNAG_ERR_ARG, "gat: null pointer");
    return SNAG_OK;
}

extern "C" SNAG_API int64_t snag_gat_work_floats(int64_t N, int f, int H) { return (int64_t)kGatBlocks * H * f * 4 + 3 * N * H + N * f + 64; }

extern "C" SNAG_API int snag_gat_layer_fwd(const float* x, int64_t N, int f, int H, const float* w, const float* a_src_dst, const int32_t* rowptr,
                                           const int32_t* col, int apply_elu, float* s_src, float* s_dst, float* agg, float* rsum, float* pre,
                                           float* out, const uint8_t* edge_keep, const int32_t* eid, int64_t n_edges, float edge_keep_scale, void* stream) {
    int rc = gat_check(N, f, H, x, w, a_src_dst, rowptr, col);
    if (rc || N == 0) return rc;
    SNAG_REQUIRE(s_src && s_dst && agg && rsum && out, SNAG_ERR_ARG, "gat_layer_fwd: null output");
    SNAG_REQUIRE(edge_keep == nullptr || (eid != nullptr && n_edges >= 0), SNAG_ERR_ARG, "gat_layer_fwd: the edge keep mask needs the CSR position -> stored edge id array");
    SNAG_REQUIRE(!apply_elu || pre, SNAG_ERR_ARG, "gat_layer_fwd: the ELU layer needs the pre-activation buffer");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)ceil_div(N, 4));
    SNAG_GAT_NU(gat_node_scores_kernel, f, grid, st, x, N, f, H, w, a_src_dst, s_src, s_dst);
    SNAG_GAT_NU(gat_aggregate_fwd_kernel, f, grid, st, x, N, f, H, w, rowptr, col, (const float*)s_src, (const float*)s_dst, apply_elu, agg, rsum, pre, out,
                edge_keep, eid, n_edges, edge_keep_scale);
    return check_launch("snag_gat_layer_fwd");
}

extern "C" SNAG_API int snag_gat_layer_bwd(const float* x, int64_t N, int f, int H, const float* w, const float* a_src_dst, const int32_t* rowptr,
                                           const int32_t* col, const int32_t* t_rowptr, const int32_t* t_col, const float* s_src,
                                           const float* s_dst, const float* agg, const float* rsum, const float* pre, int had_elu,
                                           const float* d_out, float* d_x, float* d_w, float* d_a, int accumulate, float* work,
                                           const uint8_t* edge_keep, const int32_t* eid, const int32_t* t_eid, int64_t n_edges, float edge_keep_scale,
                                           void* stream) {
    int rc = gat_check(N, f, H, x, w, a_src_dst, rowptr, col);
    if (rc || N == 0) return rc;
    SNAG_REQUIRE(s_src && s_dst && agg && rsum && d_out && d_x && d_w && d_a && work, SNAG_ERR_ARG, "gat_layer_bwd: null pointer");
    SNAG_REQUIRE(edge_keep == nullptr || (eid != nullptr && t_eid != nullptr && n_edges >= 0), SNAG_ERR_ARG,
                 "gat_layer_bwd: the edge keep mask needs both CSR position -> stored edge id arrays");
    SNAG_REQUIRE(t_rowptr && t_col, SNAG_ERR_ARG, "gat_layer_bwd: the destination-grouped edge list (t_rowptr, t_col) is required");
    SNAG_REQUIRE(!had_elu || pre, SNAG_ERR_ARG, "gat_layer_bwd: the ELU layer needs the saved pre-activation");
    hipStream_t st = (hipStream_t)stream;
    float* ds_src = work;
    float* ds_dst = ds_src + N * H;
    float* drs = ds_dst + N * H;
    float* gq = drs + N * H;
    float* part = gq + N * f;
    float* part2 = part + (int64_t)kGatBlocks * H * f;
    const int nb = (int)std::min<int64_t>(kGatBlocks, (N + 3) / 4);
    SNAG_GAT_NU(gat_bwd_out_edges_kernel, f, dim3(nb), st, x, N, f, H, w, rowptr, col, s_src, s_dst, agg, rsum, pre, had_elu, d_out, gq, drs, ds_src, part, edge_keep, eid,
                n_edges, edge_keep_scale);
    SNAG_GAT_NU(gat_bwd_in_edges_kernel, f, dim3((unsigned)ceil_div(N, 4)), st, x, N, f, H, w, t_rowptr, t_col, s_src, s_dst, rsum, (const float*)gq,
                (const float*)drs, d_x, ds_dst, edge_keep, t_eid, n_edges, edge_keep_scale);
    SNAG_GAT_NU(gat_node_scores_bwd_kernel, f, dim3(nb), st, x, N, f, H, w, a_src_dst, (const float*)ds_src, (const float*)ds_dst, d_x, part2);
    hipLaunchKernelGGL(gat_param_reduce_kernel, dim3(ceil_div(H * f, 16)), dim3(256), 0, st, (const float*)part, (const float*)part2, nb, nb, H, f, d_w, d_a, accumulate);
    return check_launch("snag_gat_layer_bwd");
}

extern "C" SNAG_API int snag_keep_scale(const float* x, int64_t n, const uint8_t* keep, float scale, float* out, void* stream) {
    SNAG_REQUIRE(n >= 0, SNAG_ERR_ARG, "keep_scale: n=%ld", (long)n);
    if (n == 0) return SNAG_OK;
    SNAG_REQUIRE(x && keep && out, SNAG_ERR_ARG, "keep_scale: null pointer");
    hipLaunchKernelGGL(keep_scale_kernel, dim3((unsigned)std::min<int64_t>(4096, (n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, n, keep, scale, out);
    return check_launch("snag_keep_scale");
}

extern "C" SNAG_API int snag_spmm_csr(const int32_t* rowptr, const int32_t* col, const float* val, const float* x, int64_t N, int f,
                                      const float* bias, float* out, void* stream) {
    SNAG_REQUIRE(N >= 0 && f >= 1 && f <= 512, SNAG_ERR_ARG, "spmm_csr: bad sizes N=%ld f=%d (f <= 512)", (long)N, f);
    if (N == 0) return SNAG_OK;
    SNAG_REQUIRE(rowptr && x && out, SNAG_ERR_ARG, "spmm_csr: null pointer");
    hipStream_t st = (hipStream_t)stream;
    SNAG_GAT_NU(spmm_csr_kernel, f, dim3((unsigned)ceil_div(N, 4)), st, rowptr, col, val, x, N, f, bias, out);
    return check_launch("snag_spmm_csr");
}
