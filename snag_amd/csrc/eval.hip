// a8: entity x entity similarity evaluation.
// reference: pairwise_distances src/utils.py:202-218, csls_sim :417-435, Runner._test main.py:376-436,
// SNAG.Iter_new_links model/SNAG.py:192-208.
//
//   Dist[i,j] = max(|x_i|^2 + |y_j|^2 - 2 x_i.y_j, 0)            (the one MFMA GEMM of the path)
//   CSLS:  r1[i] = mean top-k_j S[i,:], r2[j] = mean top-k_i S[:,j], S = 1 - Dist,
//          Dist'[i,j] = 1 - ((2 S[i,j] - r1[i]) - r2[j])            (same operation order as the reference)
//   rank_l2r[i] = #{j : Dist'[i,j] < Dist'[i,i]} ; rank_r2l[j] = #{i : Dist'[i,j] < Dist'[j,j]}
//
// The distance matrix is stored once (n x m fp32; 441 MB at the DBP15K shape is nothing on 288 GB) and
// swept twice (top-k pass, rank pass); every sweep reads whole rows with lane = column.
#include <algorithm>

#include "gemm.hip.h"

namespace snag {

constexpr int kMaxK = 16;     // csls_k: config.py:65 default 10, run_snag.sh:17 uses 3

// ---- |x_i|^2 -----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void row_sqnorm_kernel(const float* __restrict__ x, int64_t n, int d, int64_t ld, float* __restrict__ out) {
    const int64_t row = (int64_t)blockIdx.x * 4 + wave_id_uniform();
    if (row >= n) return;
    const int lane = threadIdx.x & 63;
    float s = 0.f;
    for (int c = lane; c < d; c += 64) {
        const float v = x[row * ld + c];
        s = fmaf(v, v, s);
    }
    s = sum64(s);
    if (lane == 0) out[row] = s;
}

// ---- distance epilogue of the similarity GEMM ------------------------------------------------------
struct DistEpilogue {
    float* out;
    int64_t ld;
    const float* xn;
    const float* yn;
    int M, N;
    int as_similarity;    // store 1 - dist instead of dist
    int vec;              // dwordx4 stores legal (out and ld 16-byte aligned)
    __device__ __forceinline__ float one(int m, int n, float acc) const {
        float v = fmaxf((xn[m] + yn[n]) - 2.0f * acc, 0.f);      // src/utils.py:217-218
        return as_similarity ? 1.0f - v : v;
    }
    __device__ __forceinline__ void operator()(int m, int n, float4 v) const {
        if (m >= M || n >= N) return;
        float* o = out + (int64_t)m * ld + n;
        if (vec && n + 3 < N) {
            *reinterpret_cast<float4*>(o) = make_float4(one(m, n, v.x), one(m, n + 1, v.y), one(m, n + 2, v.z), one(m, n + 3, v.w));
        } else {
            o[0] = one(m, n, v.x);
            if (n + 1 < N) o[1] = one(m, n + 1, v.y);
            if (n + 2 < N) o[2] = one(m, n + 2, v.z);
            if (n + 3 < N) o[3] = one(m, n + 3, v.w);
        }
    }
};

// ---- small sorted top-k buffer in registers ----------------------------------------------------------
// KK >= k slots (compile-time: 3, 5, 10 or 16, so that csls_k = 3 pays for 3 compare-exchanges, not 16); the k largest are v[0..k-1].
template <int KK>
struct TopK {
    float v[KK];        // descending
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int i = 0; i < KK; ++i) v[i] = -INFINITY;
    }
    __device__ __forceinline__ void push(float x) {
        if constexpr (KK <= 5) {
            // branch-free, no dependent chain: new v[0] = max(v[0], x), new v[i] = median(old v[i-1], x, old v[i]) -- x above old v[i-1]: the
            // slot takes its neighbour (shift); between the two: x itself; below: unchanged.  KK independent VALU operations per value (the
            // compare-exchange chain below is 2 KK dependent ones behind a divergent branch that a wave takes whenever ANY lane inserts).
            float p = v[0];
            v[0] = fmaxf(p, x);
#pragma unroll
            for (int i = 1; i < KK; ++i) {
                const float c = v[i];
                v[i] = __builtin_amdgcn_fmed3f(p, x, c);
                p = c;
            }
        } else {
            if (!(x > v[KK - 1])) return;       // cannot enter the buffer
#pragma unroll
            for (int i = 0; i < KK; ++i) {      // insertion by compare-exchange down the array
                const float hi = fmaxf(v[i], x), lo = fminf(v[i], x);
                v[i] = hi;
                x = lo;
            }
        }
    }
};

// mean of the k largest entries of each row of S (or of 1 - D when from_dist): one block per row
template <int KK>
__global__ __launch_bounds__(256) void row_topk_mean_kernel(const float* __restrict__ mat, int64_t ld, int n, int m, int k, int from_dist,
                                                            float* __restrict__ out, int vec) {
    __shared__ float cand[256 * KK];
    __shared__ float wmax[4];
    __shared__ int wpos[4];
    const int row = blockIdx.x;
    TopK<KK> t;
    t.init();
    const int m4 = vec ? m >> 2 : 0;            // vec: rows are 16-byte addressable -> one dwordx4 per lane (a top-k does not care about the visiting order)
    for (int q = threadIdx.x; q < m4; q += 256) {
        const float4 v = reinterpret_cast<const float4*>(mat + (int64_t)row * ld)[q];
        t.push(from_dist ? 1.0f - v.x : v.x);
        t.push(from_dist ? 1.0f - v.y : v.y);
        t.push(from_dist ? 1.0f - v.z : v.z);
        t.push(from_dist ? 1.0f - v.w : v.w);
    }
    for (int c = 4 * m4 + threadIdx.x; c < m; c += 256) {
        float s = mat[(int64_t)row * ld + c];
        if (from_dist) s = 1.0f - s;
        t.push(s);
    }
#pragma unroll
    for (int i = 0; i < KK; ++i)
        if (i < k) cand[threadIdx.x * k + i] = t.v[i];
    __syncthreads();
    // k rounds of block-wide arg-max over the 256*k candidates (ties: any equal value gives the same sum)
    const int total = 256 * k;
    float acc = 0.f;
    for (int round = 0; round < k; ++round) {
        float best = -INFINITY;
        int pos = -1;
        for (int i = threadIdx.x; i < total; i += 256) {
            const float c = cand[i];
            if (c > best) { best = c; pos = i; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ob = __shfl_xor(best, off);
            const int op = __shfl_xor(pos, off);
            if (ob > best || (ob == best && op >= 0 && (pos < 0 || op < pos))) { best = ob; pos = op; }
        }
        const int wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { wmax[wave] = best; wpos[wave] = pos; }
        __syncthreads();
        float b = wmax[0];
        int p = wpos[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (wmax[w] > b || (wmax[w] == b && wpos[w] >= 0 && (p < 0 || wpos[w] < p))) { b = wmax[w]; p = wpos[w]; }
        acc += b;                                   // descending order, like sum(topk(...)[0])
        __syncthreads();
        if (threadIdx.x == 0 && p >= 0) cand[p] = -INFINITY;
        __syncthreads();
    }
    if (threadIdx.x == 0) out[row] = acc / (float)k;
}

// per-column partial top-k over a chunk of rows: part[chunk][i][col], i < k
template <int KK>
__global__ __launch_bounds__(256) void col_topk_part_kernel(const float* __restrict__ mat, int64_t ld, int n, int m, int k, int from_dist,
                                                            int rows_per_chunk, float* __restrict__ part) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= m) return;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(n, r0 + rows_per_chunk);
    TopK<KK> t;
    t.init();
    for (int r = r0; r < r1; ++r) {
        float s = mat[(int64_t)r * ld + col];
        if (from_dist) s = 1.0f - s;
        t.push(s);
    }
#pragma unroll
    for (int i = 0; i < KK; ++i)
        if (i < k) part[((int64_t)blockIdx.y * k + i) * m + col] = t.v[i];
}

// the same with four adjacent columns per lane (dwordx4 per row; rows 16-byte addressable) and four rows requested before the first is
// consumed; one wave per block, so a block still covers 256 columns
template <int KK>
__global__ __launch_bounds__(64) void col_topk_part4_kernel(const float* __restrict__ mat, int64_t ld, int n, int m, int k, int from_dist,
                                                            int rows_per_chunk, float* __restrict__ part) {
    const int col = (blockIdx.x * 64 + threadIdx.x) * 4;
    if (col >= m) return;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(n, r0 + rows_per_chunk);
    const int nc = min(4, m - col);
    TopK<KK> t[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) t[j].init();
    auto push4 = [&](float4 v) {
        t[0].push(from_dist ? 1.0f - v.x : v.x);
        t[1].push(from_dist ? 1.0f - v.y : v.y);
        t[2].push(from_dist ? 1.0f - v.z : v.z);
        t[3].push(from_dist ? 1.0f - v.w : v.w);
    };
    if (nc == 4) {
        int r = r0;
        for (; r + 4 <= r1; r += 4) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(mat + (int64_t)(r + u) * ld + col);
#pragma unroll
            for (int u = 0; u < 4; ++u) push4(v[u]);
        }
        for (; r < r1; ++r) push4(*reinterpret_cast<const float4*>(mat + (int64_t)r * ld + col));
    } else {                                    // the last lane of a row whose length is not a multiple of 4
        for (int r = r0; r < r1; ++r)
            for (int j = 0; j < nc; ++j) {
                const float x = mat[(int64_t)r * ld + col + j];
                t[j].push(from_dist ? 1.0f - x : x);
            }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < KK; ++i)
            if (i < k && j < nc) part[((int64_t)blockIdx.y * k + i) * m + col + j] = t[j].v[i];
}

// Row AND column candidates from ONE sweep of the matrix (round 5; the row means used to be a sweep of their own: 441 MB at 10,500^2).
// One wave per (256 columns, chunk of rows), EIGHT rows at a time, the next eight in flight (a row's 1 KB per load instruction); the
// column side pushes them into its four lists as col_topk_part4_kernel does; for the row side the 8 x 256 block changes hands through
// LDS -- lane (row = lane / 8, part = lane % 8) reads 32 columns of ITS row, so the pushes are lane-local too --, the eight parts of a
// row merge with three exchanges and the row's k best of these 256 columns go to row_part[block][i][row].
// A top-k is a set: the same values as row_topk_mean_kernel / col_topk_part4_kernel collect, whatever the visiting order.
// (9 KB of LDS and 64 registers of row data per wave: a launch of 42 x 64 waves is resident at once; sixteen rows per step -- 17 KB --
// left a second, nearly empty round of waves: 130 us against the two separate sweeps' 200.)
constexpr int kRcRows = 8;
constexpr int kRcPitch = 288;           // floats per staged row: 256 + a 4-float skew per 32 columns, rows 8 bank groups apart (conflict-free reads)
__device__ __forceinline__ int rc_at(int row, int c) { return row * kRcPitch + c + 4 * (c >> 5); }
template <int KK>
__global__ __launch_bounds__(64) void rowcol_topk_part_kernel(const float* __restrict__ mat, int64_t ld, int n, int m, int k, int from_dist,
                                                              int rows_per_chunk, float* __restrict__ col_part, float* __restrict__ row_part) {
    __shared__ __attribute__((aligned(16))) float tile[kRcRows * kRcPitch];
    const int lane = threadIdx.x;
    const int col = (blockIdx.x * 64 + lane) * 4;           // (m % 4 == 0: a lane's four columns are all inside or all outside)
    const bool col_on = col < m;
    const int r0 = blockIdx.y * rows_per_chunk, r_end = min(n, r0 + rows_per_chunk);
    const int rr = lane >> 3, part = lane & 7;
    TopK<KK> tc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) tc[j].init();
    auto fetch = [&](int rb, float4 (&v)[kRcRows]) {
#pragma unroll
        for (int j = 0; j < kRcRows; ++j)                    // (clamped row and column: nothing conditional around a load)
            v[j] = *reinterpret_cast<const float4*>(mat + (int64_t)min(rb + j, r_end - 1) * ld + (col_on ? col : 0));
    };
    float4 v[kRcRows], nx[kRcRows];
    fetch(r0, nx);
    for (int rb = r0; rb < r_end; rb += kRcRows) {
#pragma unroll
        for (int j = 0; j < kRcRows; ++j) v[j] = nx[j];
        fetch(min(rb + kRcRows, r_end - 1), nx);             // the next rows travel while these are processed
#pragma unroll
        for (int j = 0; j < kRcRows; ++j) {
            if (from_dist) v[j] = make_float4(1.0f - v[j].x, 1.0f - v[j].y, 1.0f - v[j].z, 1.0f - v[j].w);
            if (!col_on) v[j] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            *reinterpret_cast<float4*>(&tile[rc_at(j, 4 * lane)]) = v[j];
        }
#pragma unroll
        for (int j = 0; j < kRcRows; ++j)
            if (rb + j < r_end) {                            // (uniform)
                tc[0].push(v[j].x); tc[1].push(v[j].y); tc[2].push(v[j].z); tc[3].push(v[j].w);
            }
        __syncthreads();
        TopK<KK> tr;
        tr.init();
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float4 w = *reinterpret_cast<const float4*>(&tile[rc_at(rr, 32 * part + 4 * t)]);
            tr.push(w.x); tr.push(w.y); tr.push(w.z); tr.push(w.w);
        }
        __syncthreads();                                     // the next rows overwrite the tile
#pragma unroll
        for (int off = 1; off <= 4; off <<= 1) {
            float o[KK];
#pragma unroll
            for (int i = 0; i < KK; ++i) o[i] = __shfl_xor(tr.v[i], off);
#pragma unroll
            for (int i = 0; i < KK; ++i) tr.push(o[i]);
        }
        if (part == 0 && rb + rr < r_end) {
#pragma unroll
            for (int i = 0; i < KK; ++i)
                if (i < k) row_part[((int64_t)blockIdx.x * k + i) * n + rb + rr] = tr.v[i];
        }
    }
    if (col_on) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < KK; ++i)
                if (i < k) col_part[((int64_t)blockIdx.y * k + i) * m + col + j] = tc[j].v[i];
    }
}

// 64 columns per block; the block's four waves merge every fourth chunk each, wave 0 then merges the four k-lists (a top-k is the same
// set whatever the order; the mean adds its values in descending order)
template <int KK>
__device__ __forceinline__ void col_topk_merge_body(const float* __restrict__ part, int chunks, int m, int k, float* __restrict__ out, int block) {
    __shared__ float ex[3][KK][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int col = block * 64 + lane;
    TopK<KK> t;
    t.init();
    if (col < m)
        for (int c = g; c < chunks; c += 4)
            for (int i = 0; i < k; ++i) t.push(part[((int64_t)c * k + i) * m + col]);
    if (g > 0) {
#pragma unroll
        for (int i = 0; i < KK; ++i) ex[g - 1][i][lane] = t.v[i];
    }
    __syncthreads();
    if (g != 0 || col >= m) return;
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
        for (int i = 0; i < KK; ++i) t.push(ex[w][i][lane]);
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < KK; ++i)
        if (i < k) acc += t.v[i];
    out[col] = acc / (float)k;
}
template <int KK>
__global__ __launch_bounds__(256) void col_topk_merge_kernel(const float* __restrict__ part, int chunks, int m, int k, float* __restrict__ out) {
    col_topk_merge_body<KK>(part, chunks, m, k, out, blockIdx.x);
}
// two candidate sets (the column side and the row side of rowcol_topk_part_kernel) finished by one launch
template <int KK>
__global__ __launch_bounds__(256) void col_topk_merge2_kernel(const float* __restrict__ part0, int chunks0, int m0, float* __restrict__ out0, int blocks0,
                                                              const float* __restrict__ part1, int chunks1, int m1, float* __restrict__ out1, int k) {
    if ((int)blockIdx.x < blocks0) col_topk_merge_body<KK>(part0, chunks0, m0, k, out0, blockIdx.x);
    else col_topk_merge_body<KK>(part1, chunks1, m1, k, out1, blockIdx.x - blocks0);
}

// the value the ranks compare: Dist' (CSLS) or Dist
__device__ __forceinline__ float eval_dist(float d, float r1, float r2, int csls) {
    if (!csls) return d;
    const float s = 1.0f - d;                          // main.py:392  (1 - distance)
    return 1.0f - ((2.0f * s - r1) - r2);              // utils.py:433-434 then main.py:392
}

// out[i,j] = 2 S[i,j] - r1[i] - r2[j]   (csls_sim as a free function, src/utils.py:417-435)
__global__ __launch_bounds__(256) void csls_apply_kernel(const float* __restrict__ sim, int64_t ld, int n, int m, const float* __restrict__ r1,
                                                         const float* __restrict__ r2, float* __restrict__ out) {
    const int64_t total = (int64_t)n * m;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int r = (int)(i / m), c = (int)(i % m);
        out[i] = (2.0f * sim[(int64_t)r * ld + c] - r1[r]) - r2[c];
    }
}

// this thread's share of #{c : D'[row, c] < ref}; vec: the row and r2 are 16-byte addressable (a count does not care about the visiting order)
__device__ __forceinline__ int row_count_less(const float* __restrict__ row, int m, float a, const float* __restrict__ r2, int csls, float ref, int vec) {
    int cnt = 0;
    const int m4 = vec ? m >> 2 : 0;
    for (int q = threadIdx.x; q < m4; q += 256) {
        const float4 d = reinterpret_cast<const float4*>(row)[q];
        const float4 b = csls ? reinterpret_cast<const float4*>(r2)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        cnt += (eval_dist(d.x, a, b.x, csls) < ref ? 1 : 0) + (eval_dist(d.y, a, b.y, csls) < ref ? 1 : 0) + (eval_dist(d.z, a, b.z, csls) < ref ? 1 : 0) +
               (eval_dist(d.w, a, b.w, csls) < ref ? 1 : 0);
    }
    for (int c = 4 * m4 + threadIdx.x; c < m; c += 256) cnt += eval_dist(row[c], a, csls ? r2[c] : 0.f, csls) < ref ? 1 : 0;
    return cnt;
}

// rank_l2r[i] = #{j : D'[i,j] < D'[i,i]} : one block per row
__global__ __launch_bounds__(256) void rank_rows_kernel(const float* __restrict__ dist, int64_t ld, int n, int m, const float* __restrict__ r1,
                                                        const float* __restrict__ r2, int csls, int* __restrict__ rank, int vec) {
    __shared__ int red[4];
    const int row = blockIdx.x;
    if (row >= m) {                                   // no diagonal element for this row
        if (threadIdx.x == 0) rank[row] = -1;
        return;
    }
    const float a = csls ? r1[row] : 0.f;
    const float ref = eval_dist(dist[(int64_t)row * ld + row], a, csls ? r2[row] : 0.f, csls);
    int cnt = row_count_less(dist + (int64_t)row * ld, m, a, r2, csls, ref, vec);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) rank[row] = red[0] + red[1] + red[2] + red[3];
}

// rank_r2l[j] = #{i : D'[i,j] < D'[j,j]} : thread per column, rows in chunks, integer atomics (exact)
__global__ __launch_bounds__(256) void rank_cols_kernel(const float* __restrict__ dist, int64_t ld, int n, int m, const float* __restrict__ r1,
                                                        const float* __restrict__ r2, int csls, int rows_per_chunk, int* __restrict__ rank) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= m || col >= n) return;
    const float b = csls ? r2[col] : 0.f;
    const float ref = eval_dist(dist[(int64_t)col * ld + col], csls ? r1[col] : 0.f, b, csls);
    const int r0 = blockIdx.y * rows_per_chunk, r1e = min(n, r0 + rows_per_chunk);
    int cnt = 0;
    for (int r = r0; r < r1e; ++r) cnt += eval_dist(dist[(int64_t)r * ld + col], csls ? r1[r] : 0.f, b, csls) < ref ? 1 : 0;
    atomicAdd(rank + col, cnt);
}

// the same with four adjacent columns per lane (dwordx4 per row) and four rows requested at a time; one wave per block
// (DIAG: the reference values come gathered in diag[] -- the row-sharded evaluation, where the diagonal element of a column lives on another rank)
template <bool DIAG>
__global__ __launch_bounds__(64) void rank_cols4_kernel(const float* __restrict__ dist, int64_t ld, int n, int m, const float* __restrict__ r1,
                                                        const float* __restrict__ r2, int csls, const float* __restrict__ diag, int rows_per_chunk,
                                                        int* __restrict__ rank) {
    const int col = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int ncols = DIAG ? m : min(m, n);
    if (col >= ncols) return;
    const int nc = min(4, ncols - col);
    const int r0 = blockIdx.y * rows_per_chunk, r1e = min(n, r0 + rows_per_chunk);
    float b[4], ref[4];
    int cnt[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = col + min(j, nc - 1);
        b[j] = csls ? r2[c] : 0.f;
        ref[j] = DIAG ? diag[c] : eval_dist(dist[(int64_t)c * ld + c], csls ? r1[c] : 0.f, b[j], csls);
    }
    if (nc == 4) {
        auto count4 = [&](float4 d, float a) {
            cnt[0] += eval_dist(d.x, a, b[0], csls) < ref[0] ? 1 : 0;
            cnt[1] += eval_dist(d.y, a, b[1], csls) < ref[1] ? 1 : 0;
            cnt[2] += eval_dist(d.z, a, b[2], csls) < ref[2] ? 1 : 0;
            cnt[3] += eval_dist(d.w, a, b[3], csls) < ref[3] ? 1 : 0;
        };
        int r = r0;
        for (; r + 4 <= r1e; r += 4) {
            float4 d[4];
            float a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                d[u] = *reinterpret_cast<const float4*>(dist + (int64_t)(r + u) * ld + col);
                a[u] = csls ? r1[r + u] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) count4(d[u], a[u]);
        }
        for (; r < r1e; ++r) count4(*reinterpret_cast<const float4*>(dist + (int64_t)r * ld + col), csls ? r1[r] : 0.f);
    } else {
        for (int r = r0; r < r1e; ++r)
            for (int j = 0; j < nc; ++j) cnt[j] += eval_dist(dist[(int64_t)r * ld + col + j], csls ? r1[r] : 0.f, b[j], csls) < ref[j] ? 1 : 0;
    }
    for (int j = 0; j < nc; ++j) atomicAdd(rank + col + j, cnt[j]);
}

// BOTH directions in ONE sweep of the square matrix (the separate row and column passes read its 441 MB twice at 10,500^2): the column
// kernel's walk -- a lane owns four adjacent columns, a wave 256, the workgroup a chunk of rows -- with the row counts taken on the way: for
// row r the wave's 256 values are compared with the row's own reference, four ballots count the hits and one integer atomic adds them to
// rank_l2r[r].  The same eval_dist values feed both comparisons; integer counts do not care who adds what.  n == m, dwordx4-addressable rows.
__global__ __launch_bounds__(64) void rank_both4_kernel(const float* __restrict__ dist, int64_t ld, int n, const float* __restrict__ r1,
                                                        const float* __restrict__ r2, int csls, int rows_per_chunk, int* __restrict__ rank_rows,
                                                        int* __restrict__ rank_cols) {
    // (n % 4 == 0: a lane owns four whole columns or none; a lane past the edge walks the last four columns again and counts nothing)
    const int col_raw = (blockIdx.x * 64 + threadIdx.x) * 4;
    const bool live = col_raw < n;
    const int col = live ? col_raw : n - 4;
    const int r0 = blockIdx.y * rows_per_chunk, r1e = min(n, r0 + rows_per_chunk);
    float b[4], ref[4];
    int cnt[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = col + j;
        b[j] = csls ? r2[c] : 0.f;
        ref[j] = eval_dist(dist[(int64_t)c * ld + c], csls ? r1[c] : 0.f, b[j], csls);
    }
    // this lane's hits of one row (0..4): column counts as a side effect
    auto lane_hits = [&](float4 d, float a, float rref) {
        const float v0 = eval_dist(d.x, a, b[0], csls), v1 = eval_dist(d.y, a, b[1], csls), v2 = eval_dist(d.z, a, b[2], csls), v3 = eval_dist(d.w, a, b[3], csls);
        cnt[0] += v0 < ref[0] ? 1 : 0;
        cnt[1] += v1 < ref[1] ? 1 : 0;
        cnt[2] += v2 < ref[2] ? 1 : 0;
        cnt[3] += v3 < ref[3] ? 1 : 0;
        const int hits = (v0 < rref ? 1 : 0) + (v1 < rref ? 1 : 0) + (v2 < rref ? 1 : 0) + (v3 < rref ? 1 : 0);
        return live ? hits : 0;
    };
    auto one_row = [&](int r, float4 d, float a, float rref) {
        const int hits = isum64(lane_hits(d, a, rref));
        if (threadIdx.x == 0 && hits != 0) atomicAdd(rank_rows + r, hits);
    };
    auto load_row = [&](int r) { return *reinterpret_cast<const float4*>(dist + (int64_t)r * ld + col); };
    // 64 rows at a time: lane i fetches row s0 + i's neighbourhood term and its own diagonal element ONCE (three loads per lane and 64
    // rows, not three wave-wide loads per row); a row's two scalars then come out of that lane's registers
    for (int s0 = r0; s0 < r1e; s0 += 64) {
        const int mine = min(s0 + (int)threadIdx.x, r1e - 1);
        const float a_l = csls ? r1[mine] : 0.f;
        const float rr_l = eval_dist(dist[(int64_t)mine * ld + mine], a_l, csls ? r2[mine] : 0.f, csls);
        const int cnt_rows = min(64, r1e - s0);
        int i = 0;
        for (; i + 4 <= cnt_rows; i += 4) {
            float4 d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) d[u] = load_row(s0 + i + u);
            // four rows' hit counts travel through TWO wave reductions, 16 bits per row (a wave's count is at most 256)
            int h[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) h[u] = lane_hits(d[u], __shfl(a_l, i + u), __shfl(rr_l, i + u));
            const int t01 = isum64(h[0] | (h[1] << 16)), t23 = isum64(h[2] | (h[3] << 16));
            if (threadIdx.x < 4) {
                const int mine_hits = threadIdx.x == 0 ? (t01 & 0xffff) : threadIdx.x == 1 ? (t01 >> 16) : threadIdx.x == 2 ? (t23 & 0xffff) : (t23 >> 16);
                if (mine_hits != 0) atomicAdd(rank_rows + s0 + i + (int)threadIdx.x, mine_hits);
            }
        }
        for (; i < cnt_rows; ++i) one_row(s0 + i, load_row(s0 + i), __shfl(a_l, i), __shfl(rr_l, i));
    }
    if (live)
        for (int j = 0; j < 4; ++j) atomicAdd(rank_cols + col + j, cnt[j]);
}

// the k largest values per column (descending) instead of their mean: what a rank contributes to the global column neighbourhoods
template <int KK>
__global__ __launch_bounds__(256) void col_topk_values_kernel(const float* __restrict__ part, int chunks, int m, int k, float* __restrict__ out) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= m) return;
    TopK<KK> t;
    t.init();
    for (int c = 0; c < chunks; ++c)
        for (int i = 0; i < k; ++i) t.push(part[((int64_t)c * k + i) * m + col]);
#pragma unroll
    for (int i = 0; i < KK; ++i)
        if (i < k) out[(int64_t)i * m + col] = t.v[i];
}

// Row-sharded evaluation (SURVEY par. 8e, config 4): a rank holds rows [row0, row0 + n) of the distance matrix, all m columns.
// rank_l2r of its rows and the CSLS-adjusted diagonal values D'[row0+i, row0+i] ...
__global__ __launch_bounds__(256) void rank_rows_block_kernel(const float* __restrict__ dist, int64_t ld, int n, int m, int row0,
                                                              const float* __restrict__ r1, const float* __restrict__ r2, int csls,
                                                              int* __restrict__ rank, float* __restrict__ diag, int vec) {
    __shared__ int red[4];
    const int row = blockIdx.x, dcol = row0 + row;
    const float a = csls ? r1[row] : 0.f;
    const float ref = eval_dist(dist[(int64_t)row * ld + dcol], a, csls ? r2[dcol] : 0.f, csls);
    int cnt = row_count_less(dist + (int64_t)row * ld, m, a, r2, csls, ref, vec);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        rank[row] = red[0] + red[1] + red[2] + red[3];
        diag[row] = ref;
    }
}
// ... and this rank's share of rank_r2l: count[j] += #{local i : D'[i,j] < diag[j]} with the gathered diagonal (integer, exact)
__global__ __launch_bounds__(256) void rank_cols_block_kernel(const float* __restrict__ dist, int64_t ld, int n, int m, const float* __restrict__ r1,
                                                              const float* __restrict__ r2, int csls, const float* __restrict__ diag,
                                                              int rows_per_chunk, int* __restrict__ count) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= m) return;
    const float b = csls ? r2[col] : 0.f, ref = diag[col];
    const int r0 = blockIdx.y * rows_per_chunk, r1e = min(n, r0 + rows_per_chunk);
    int cnt = 0;
    for (int r = r0; r < r1e; ++r) cnt += eval_dist(dist[(int64_t)r * ld + col], csls ? r1[r] : 0.f, b, csls) < ref ? 1 : 0;
    atomicAdd(count + col, cnt);
}

// Hits@k counts, sum(rank+1), sum(1/(rank+1)) in double, fixed order  (main.py:400-407,430-436)
__global__ __launch_bounds__(256) void rank_summary_kernel(const int* __restrict__ rank, int n, const int* __restrict__ topk, int ntop,
                                                           double* __restrict__ out, const int* __restrict__ rank_b) {
    if (blockIdx.x == 1) {                        // second direction of the same evaluation (snag_rank_summary_pair): block 1, out + ntop + 2
        rank = rank_b;
        out += ntop + 2;
    }
    // one pass over the ranks for all ntop + 2 sums (ntop <= 8); each sum keeps its own thread-strided order and the same tree
    __shared__ double red[256];
    double s[10];
#pragma unroll
    for (int q = 0; q < 10; ++q) s[q] = 0.0;
    int tk[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) tk[q] = q < ntop ? topk[q] : 0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int r = rank[i];
#pragma unroll
        for (int q = 0; q < 8; ++q) s[q] += (q < ntop && r < tk[q]) ? 1.0 : 0.0;
        s[8] += (double)(r + 1);
        s[9] += 1.0 / (double)(r + 1);
    }
#pragma unroll
    for (int q = 0; q < 10; ++q) {
        if (q < 8 && q >= ntop) continue;
        red[threadIdx.x] = s[q];
        __syncthreads();
        for (int off = 128; off >= 1; off >>= 1) {
            if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[q < 8 ? q : ntop + (q - 8)] = red[0];
        __syncthreads();
    }
}

// ---- arg-min per row / per column (Iter_new_links, model/SNAG.py:200-201); ties -> smallest index ----
__global__ __launch_bounds__(256) void argmin_rows_kernel(const float* __restrict__ dist, int64_t ld, int n, int m, int* __restrict__ out, int vec) {
    __shared__ float bv[4];
    __shared__ int bi[4];
    const int row = blockIdx.x;
    float best = INFINITY;
    int pos = 0x7fffffff;
    const int m4 = vec ? m >> 2 : 0;            // vec: 16-byte addressable rows; a thread still meets its columns in ascending order
    for (int q = threadIdx.x; q < m4; q += 256) {
        const float4 v = reinterpret_cast<const float4*>(dist + (int64_t)row * ld)[q];
        if (v.x < best) { best = v.x; pos = 4 * q; }
        if (v.y < best) { best = v.y; pos = 4 * q + 1; }
        if (v.z < best) { best = v.z; pos = 4 * q + 2; }
        if (v.w < best) { best = v.w; pos = 4 * q + 3; }
    }
    for (int c = 4 * m4 + threadIdx.x; c < m; c += 256) {
        const float v = dist[(int64_t)row * ld + c];
        if (v < best || (v == best && c < pos)) { best = v; pos = c; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ob = __shfl_xor(best, off);
        const int op = __shfl_xor(pos, off);
        if (ob < best || (ob == best && op < pos)) { best = ob; pos = op; }
    }
    if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = pos; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (bv[w] < best || (bv[w] == best && bi[w] < pos)) { best = bv[w]; pos = bi[w]; }
        out[row] = pos;
    }
}

__global__ __launch_bounds__(256) void argmin_cols_part_kernel(const float* __restrict__ dist, int64_t ld, int n, int m, int rows_per_chunk,
                                                               float* __restrict__ pv, int* __restrict__ pi) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= m) return;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(n, r0 + rows_per_chunk);
    float best = INFINITY;
    int pos = 0x7fffffff;
    for (int r = r0; r < r1; ++r) {
        const float v = dist[(int64_t)r * ld + col];
        if (v < best) { best = v; pos = r; }
    }
    pv[(int64_t)blockIdx.y * m + col] = best;
    pi[(int64_t)blockIdx.y * m + col] = pos;
}

// four adjacent columns per lane (dwordx4 per row, four rows requested at a time); one wave per block; rows ascending, so the first of equals stays
__global__ __launch_bounds__(64) void argmin_cols_part4_kernel(const float* __restrict__ dist, int64_t ld, int n, int m, int rows_per_chunk,
                                                               float* __restrict__ pv, int* __restrict__ pi) {
    const int col = (blockIdx.x * 64 + threadIdx.x) * 4;
    if (col >= m) return;
    const int nc = min(4, m - col);
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(n, r0 + rows_per_chunk);
    float best[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
    int pos[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
    if (nc == 4) {
        auto take = [&](float4 v, int r) {
            if (v.x < best[0]) { best[0] = v.x; pos[0] = r; }
            if (v.y < best[1]) { best[1] = v.y; pos[1] = r; }
            if (v.z < best[2]) { best[2] = v.z; pos[2] = r; }
            if (v.w < best[3]) { best[3] = v.w; pos[3] = r; }
        };
        int r = r0;
        for (; r + 4 <= r1; r += 4) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(dist + (int64_t)(r + u) * ld + col);
#pragma unroll
            for (int u = 0; u < 4; ++u) take(v[u], r + u);
        }
        for (; r < r1; ++r) take(*reinterpret_cast<const float4*>(dist + (int64_t)r * ld + col), r);
    } else {
        for (int r = r0; r < r1; ++r)
            for (int j = 0; j < nc; ++j) {
                const float v = dist[(int64_t)r * ld + col + j];
                if (v < best[j]) { best[j] = v; pos[j] = r; }
            }
    }
    for (int j = 0; j < nc; ++j) {
        pv[(int64_t)blockIdx.y * m + col + j] = best[j];
        pi[(int64_t)blockIdx.y * m + col + j] = pos[j];
    }
}

__global__ __launch_bounds__(256) void argmin_cols_merge_kernel(const float* __restrict__ pv, const int* __restrict__ pi, int chunks, int m,
                                                                int* __restrict__ out) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= m) return;
    float best = INFINITY;
    int pos = 0x7fffffff;
    for (int c = 0; c < chunks; ++c) {
        const float v = pv[(int64_t)c * m + col];
        const int p = pi[(int64_t)c * m + col];
        if (v < best || (v == best && p < pos)) { best = v; pos = p; }
    }
    out[col] = pos;
}

static constexpr int kColChunks = 64;
// rows of the matrix are 16-byte addressable: the dwordx4 forms of the row / column passes apply
static int matrix_vec_ok(const float* mat, int64_t ld) { return (ld % 4 == 0) && (((uintptr_t)mat & 15) == 0); }

static void launch_dist_gemm(const GemmOperands& p, const DistEpilogue& e, hipStream_t st) {
    const int64_t big_tiles = (int64_t)ceil_div(p.M, 128) * ceil_div(p.N, 128);
    const dim3 gbig(ceil_div(p.N, 128), ceil_div(p.M, 128), 1), gsmall(ceil_div(p.N, 128), ceil_div(p.M, 64), 1);
    // Large problems: the eight-wave row-private 256 x 128 tile (A straight to registers, B through the LDS, two wave groups half an
    // iteration apart).  The kernel is bound by what the CUs pull in per tile ((BM + BN) K floats): 1.77 -> 1.52 ms against the
    // 2 x 2-wave 128 x 128 tile at 10,500 x 10,500 x 1,200 (row-private 128 x 128: 1.72).  VEC = false: the loaders pick 16-byte or
    // 4-byte loads at run time from p.a_vec / p.b_vec (the counted-wait form measured 1.65 vs 1.63 ms here).  Tiles are visited in groups
    // of 4 row tiles per XCD (1.507 -> 1.473 ms against row-major order).
    if (big_tiles >= 1024) {
        GemmOperands pg = p;
        pg.xcd = 1 | (4 << 4);
        hipLaunchKernelGGL((gemm_tile_kernel<256, 128, 32, 128, true, true, false, DistEpilogue>), dim3(ceil_div(p.N, 128), ceil_div(p.M, 256), 1), dim3(512), 0, st, pg, e,
                           (float*)nullptr);
    } else if (big_tiles >= 512) {
        hipLaunchKernelGGL((gemm_tile_kernel<128, 128, 64, 64, true, true, false, DistEpilogue>), gbig, dim3(256), 0, st, p, e, (float*)nullptr);
    } else {
        hipLaunchKernelGGL((gemm_tile_kernel<64, 128, 32, 64, true, true, false, DistEpilogue>), gsmall, dim3(256), 0, st, p, e, (float*)nullptr);
    }
}

}  // namespace snag

using namespace snag;

extern "C" SNAG_API int snag_row_sqnorm(const float* x, int64_t n, int d, int64_t ld, float* out, void* stream) {
    SNAG_REQUIRE(n >= 0 && d >= 1 && ld >= d, SNAG_ERR_ARG, "row_sqnorm: bad sizes");
    if (n == 0) return SNAG_OK;
    SNAG_REQUIRE(x && out, SNAG_ERR_ARG, "row_sqnorm: null pointer");
    hipLaunchKernelGGL(row_sqnorm_kernel, dim3((unsigned)ceil_div(n, 4)), dim3(256), 0, (hipStream_t)stream, x, n, d, ld, out);
    return check_launch("snag_row_sqnorm");
}

extern "C" SNAG_API int snag_pairwise_dist(const float* x, int64_t ldx, int n, const float* y, int64_t ldy, int m, int d, const float* xn,
                                           const float* yn, float* out, int64_t ld_out, int as_similarity, void* stream) {
    SNAG_REQUIRE(n >= 0 && m >= 0 && d >= 1 && ldx >= d && ldy >= d && ld_out >= m, SNAG_ERR_ARG, "pairwise_dist: bad sizes n=%d m=%d d=%d", n, m, d);
    if (n == 0 || m == 0) return SNAG_OK;
    SNAG_REQUIRE(x && y && xn && yn && out, SNAG_ERR_ARG, "pairwise_dist: null pointer");
    GemmOperands p;
    p.A = x; p.B = y; p.lda = ldx; p.ldb = ldy; p.M = n; p.N = m; p.K = d;
    p.a_vec = (ldx % 4 == 0) && (((uintptr_t)x & 15) == 0);
    p.b_vec = (ldy % 4 == 0) && (((uintptr_t)y & 15) == 0);
    p.k_per_split = ceil_div(d, kBK) * kBK;
    p.dbg = 0;
    p.xcd = 1;
    DistEpilogue e{out, ld_out, xn, yn, n, m, as_similarity, (int)((ld_out % 4 == 0) && (((uintptr_t)out & 15) == 0))};
    launch_dist_gemm(p, e, (hipStream_t)stream);
    return check_launch("snag_pairwise_dist");
}

namespace snag {
// out[i,j] = sum_k |x[i,k] - y[j,k]|  (--distance 1: scipy cdist(metric="cityblock") in Runner._test, main.py:388-391).
// Not a GEMM: 64 x 64 output tile per block, 4 x 4 per thread, k staged through LDS in slabs of 32 (k-major, pitch 65).
__global__ __launch_bounds__(256) void cityblock_kernel(const float* __restrict__ x, int64_t ldx, int n, const float* __restrict__ y, int64_t ldy, int m,
                                                        int d, float* __restrict__ out, int64_t ldo) {
    __shared__ float xs[32][65], ys[32][65];
    const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
    for (int k0 = 0; k0 < d; k0 += 32) {
        for (int e = threadIdx.x; e < 64 * 32; e += 256) {
            const int r = e >> 5, k = e & 31;
            xs[k][r] = (i0 + r < n && k0 + k < d) ? x[(int64_t)(i0 + r) * ldx + k0 + k] : 0.f;
            ys[k][r] = (j0 + r < m && k0 + k < d) ? y[(int64_t)(j0 + r) * ldy + k0 + k] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < 32; ++k) {
            float xv[4], yv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) xv[a] = xs[k][ty + 16 * a];
#pragma unroll
            for (int b = 0; b < 4; ++b) yv[b] = ys[k][tx + 16 * b];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] += fabsf(xv[a] - yv[b]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int i = i0 + ty + 16 * a, j = j0 + tx + 16 * b;
            if (i < n && j < m) out[(int64_t)i * ldo + j] = acc[a][b];
        }
}
}  // namespace snag

extern "C" SNAG_API int snag_cityblock_dist(const float* x, int64_t ldx, int n, const float* y, int64_t ldy, int m, int d, float* out, int64_t ld_out,
                                            void* stream) {
    SNAG_REQUIRE(n >= 0 && m >= 0 && d >= 0 && ldx >= d && ldy >= d && ld_out >= m, SNAG_ERR_ARG, "cityblock_dist: bad sizes");
    if (n == 0 || m == 0) return SNAG_OK;
    SNAG_REQUIRE(out && ((x && y) || d == 0), SNAG_ERR_ARG, "cityblock_dist: null pointer");
    SNAG_REQUIRE(ceil_div(n, 64) <= 65535, SNAG_ERR_UNSUPPORTED, "cityblock_dist: more than 4M rows");
    hipLaunchKernelGGL(cityblock_kernel, dim3(ceil_div(m, 64), ceil_div(n, 64)), dim3(256), 0, (hipStream_t)stream, x, ldx, n, y, ldy, m, d, out, ld_out);
    return check_launch("snag_cityblock_dist");
}

extern "C" SNAG_API int64_t snag_eval_work_floats(int n, int m) { return (int64_t)2 * kColChunks * kMaxK * std::max(n, m) + 64; }

extern "C" SNAG_API int snag_csls_neighbourhood(const float* mat, int64_t ld, int n, int m, int k, int from_dist, float* r1, float* r2, float* work,
                                                void* stream) {
    SNAG_REQUIRE(n >= 1 && m >= 1 && ld >= m, SNAG_ERR_ARG, "csls: bad sizes");
    SNAG_REQUIRE(k >= 1 && k <= kMaxK, SNAG_ERR_UNSUPPORTED, "csls: k=%d outside [1, %d]", k, kMaxK);
    SNAG_REQUIRE(k <= n && k <= m, SNAG_ERR_ARG, "csls: k=%d larger than the matrix (%d x %d)", k, n, m);   // torch.topk raises too
    SNAG_REQUIRE(mat && r1 && r2 && work, SNAG_ERR_ARG, "csls: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int chunks = std::min(kColChunks, n);
    const int rpc = ceil_div(n, chunks);
    const int nch = ceil_div(n, rpc);
    const int vec = matrix_vec_ok(mat, ld);
    const int cb = ceil_div(m, 256);
    if (vec && m % 4 == 0 && cb <= kColChunks) {
        // ONE sweep for both neighbourhoods: row candidates per 256-column block behind the column candidates in the workspace
        const int rpc16 = ceil_div(rpc, kRcRows) * kRcRows, nch16 = ceil_div(n, rpc16);
        float* row_part = work + (int64_t)kColChunks * kMaxK * std::max(n, m);
#define SNAG_TOPK(KK)                                                                                                                       \
    do {                                                                                                                                    \
        hipLaunchKernelGGL((rowcol_topk_part_kernel<KK>), dim3(cb, nch16), dim3(64), 0, st, mat, ld, n, m, k, from_dist, rpc16, work, row_part); \
        hipLaunchKernelGGL((col_topk_merge2_kernel<KK>), dim3(ceil_div(m, 64) + ceil_div(n, 64)), dim3(256), 0, st, (const float*)work, nch16, m, r2, \
                           ceil_div(m, 64), (const float*)row_part, cb, n, r1, k);                                                          \
    } while (0)
        if (k <= 3) SNAG_TOPK(3);
        else if (k <= 5) SNAG_TOPK(5);
        else if (k <= 10) SNAG_TOPK(10);
        else SNAG_TOPK(16);
#undef SNAG_TOPK
        return check_launch("snag_csls_neighbourhood");
    }
#define SNAG_TOPK(KK)                                                                                                                       \
    do {                                                                                                                                    \
        hipLaunchKernelGGL((row_topk_mean_kernel<KK>), dim3(n), dim3(256), 0, st, mat, ld, n, m, k, from_dist, r1, vec);                     \
        if (vec) hipLaunchKernelGGL((col_topk_part4_kernel<KK>), dim3(ceil_div(m, 256), nch), dim3(64), 0, st, mat, ld, n, m, k, from_dist, rpc, work); \
        else hipLaunchKernelGGL((col_topk_part_kernel<KK>), dim3(ceil_div(m, 256), nch), dim3(256), 0, st, mat, ld, n, m, k, from_dist, rpc, work); \
        hipLaunchKernelGGL((col_topk_merge_kernel<KK>), dim3(ceil_div(m, 64)), dim3(256), 0, st, (const float*)work, nch, m, k, r2);          \
    } while (0)
    if (k <= 3) SNAG_TOPK(3);
    else if (k <= 5) SNAG_TOPK(5);
    else if (k <= 10) SNAG_TOPK(10);
    else SNAG_TOPK(16);
#undef SNAG_TOPK
    return check_launch("snag_csls_neighbourhood");
}

extern "C" SNAG_API int snag_csls_apply(const float* sim, int64_t ld, int n, int m, const float* r1, const float* r2, float* out, void* stream) {
    SNAG_REQUIRE(n >= 0 && m >= 0 && ld >= m, SNAG_ERR_ARG, "csls_apply: bad sizes");
    if (n == 0 || m == 0) return SNAG_OK;
    SNAG_REQUIRE(sim && r1 && r2 && out, SNAG_ERR_ARG, "csls_apply: null pointer");
    const int64_t total = (int64_t)n * m;
    hipLaunchKernelGGL(csls_apply_kernel, dim3((unsigned)std::min<int64_t>(8192, (total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sim, ld,
                       n, m, r1, r2, out);
    return check_launch("snag_csls_apply");
}

extern "C" SNAG_API int snag_rank_diagonal(const float* dist, int64_t ld, int n, int m, const float* r1, const float* r2, int32_t* rank_l2r,
                                           int32_t* rank_r2l, void* stream) {
    SNAG_REQUIRE(n >= 1 && m >= 1 && ld >= m, SNAG_ERR_ARG, "rank_diagonal: bad sizes");
    SNAG_REQUIRE(n == m, SNAG_ERR_ARG, "rank_diagonal: the test pairs form a square matrix (n=%d, m=%d)", n, m);
    SNAG_REQUIRE(dist && (rank_l2r || rank_r2l), SNAG_ERR_ARG, "rank_diagonal: null pointer");
    SNAG_REQUIRE((r1 == nullptr) == (r2 == nullptr), SNAG_ERR_ARG, "rank_diagonal: r1 and r2 go together");
    hipStream_t st = (hipStream_t)stream;
    const int csls = r1 != nullptr;
    const int vec = matrix_vec_ok(dist, ld);
    if (rank_l2r && rank_r2l && vec && n % 4 == 0 && n >= 4) {
        // both directions from ONE sweep of the matrix
        if (hipMemsetAsync(rank_l2r, 0, sizeof(int32_t) * n, st) != hipSuccess || hipMemsetAsync(rank_r2l, 0, sizeof(int32_t) * m, st) != hipSuccess)
            return fail(SNAG_ERR_LAUNCH, "rank_diagonal: memset failed");
        const int chunks = std::min(kColChunks, n);
        const int rpc = ceil_div(n, chunks);
        hipLaunchKernelGGL(rank_both4_kernel, dim3(ceil_div(n, 256), ceil_div(n, rpc)), dim3(64), 0, st, dist, ld, n, r1, r2, csls, rpc, rank_l2r, rank_r2l);
        return check_launch("snag_rank_diagonal");
    }
    if (rank_l2r) hipLaunchKernelGGL(rank_rows_kernel, dim3(n), dim3(256), 0, st, dist, ld, n, m, r1, r2, csls, rank_l2r, vec && (((uintptr_t)r2 & 15) == 0));
    if (rank_r2l) {
        if (hipMemsetAsync(rank_r2l, 0, sizeof(int32_t) * m, st) != hipSuccess) return fail(SNAG_ERR_LAUNCH, "rank_diagonal: memset failed");
        const int chunks = std::min(kColChunks, n);
        const int rpc = ceil_div(n, chunks);
        if (vec) hipLaunchKernelGGL(rank_cols4_kernel<false>, dim3(ceil_div(m, 256), ceil_div(n, rpc)), dim3(64), 0, st, dist, ld, n, m, r1, r2, csls, (const float*)nullptr, rpc, rank_r2l);
        else hipLaunchKernelGGL(rank_cols_kernel, dim3(ceil_div(m, 256), ceil_div(n, rpc)), dim3(256), 0, st, dist, ld, n, m, r1, r2, csls, rpc, rank_r2l);
    }
    return check_launch("snag_rank_diagonal");
}

extern "C" SNAG_API int snag_col_topk(const float* mat, int64_t ld, int n, int m, int k, int from_dist, float* out, float* work, void* stream) {
    SNAG_REQUIRE(n >= 1 && m >= 1 && ld >= m, SNAG_ERR_ARG, "col_topk: bad sizes");
    SNAG_REQUIRE(k >= 1 && k <= kMaxK, SNAG_ERR_UNSUPPORTED, "col_topk: k=%d outside [1, %d]", k, kMaxK);
    SNAG_REQUIRE(mat && out && work, SNAG_ERR_ARG, "col_topk: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int chunks = std::min(kColChunks, n);
    const int rpc = ceil_div(n, chunks);
    const int nch = ceil_div(n, rpc);
#define SNAG_TOPK(KK)                                                                                                                       \
    do {                                                                                                                                    \
        if (matrix_vec_ok(mat, ld)) hipLaunchKernelGGL((col_topk_part4_kernel<KK>), dim3(ceil_div(m, 256), nch), dim3(64), 0, st, mat, ld, n, m, k, from_dist, rpc, work); \
        else hipLaunchKernelGGL((col_topk_part_kernel<KK>), dim3(ceil_div(m, 256), nch), dim3(256), 0, st, mat, ld, n, m, k, from_dist, rpc, work); \
        hipLaunchKernelGGL((col_topk_values_kernel<KK>), dim3(ceil_div(m, 256)), dim3(256), 0, st, (const float*)work, nch, m, k, out);       \
    } while (0)
    if (k <= 3) SNAG_TOPK(3);
    else if (k <= 5) SNAG_TOPK(5);
    else if (k <= 10) SNAG_TOPK(10);
    else SNAG_TOPK(16);
#undef SNAG_TOPK
    return check_launch("snag_col_topk");
}

extern "C" SNAG_API int snag_col_topk_mean(const float* cand, int chunks, int m, int k, float* out, void* stream) {
    SNAG_REQUIRE(cand && out && chunks >= 1 && m >= 1, SNAG_ERR_ARG, "col_topk_mean: bad arguments");
    SNAG_REQUIRE(k >= 1 && k <= kMaxK, SNAG_ERR_UNSUPPORTED, "col_topk_mean: k=%d outside [1, %d]", k, kMaxK);
    hipStream_t st = (hipStream_t)stream;
#define SNAG_TOPK(KK) hipLaunchKernelGGL((col_topk_merge_kernel<KK>), dim3(ceil_div(m, 64)), dim3(256), 0, st, cand, chunks, m, k, out)
    if (k <= 3) SNAG_TOPK(3);
    else if (k <= 5) SNAG_TOPK(5);
    else if (k <= 10) SNAG_TOPK(10);
    else SNAG_TOPK(16);
#undef SNAG_TOPK
    return check_launch("snag_col_topk_mean");
}

extern "C" SNAG_API int snag_rank_rows_block(const float* dist, int64_t ld, int n, int m, int row0, const float* r1, const float* r2, int32_t* rank,
                                             float* diag, void* stream) {
    SNAG_REQUIRE(n >= 1 && m >= 1 && ld >= m && row0 >= 0 && row0 + n <= m, SNAG_ERR_ARG, "rank_rows_block: rows [%d, %d) outside the %d columns", row0,
                 row0 + n, m);
    SNAG_REQUIRE(dist && rank && diag, SNAG_ERR_ARG, "rank_rows_block: null pointer");
    SNAG_REQUIRE((r1 == nullptr) == (r2 == nullptr), SNAG_ERR_ARG, "rank_rows_block: r1 and r2 go together");
    hipLaunchKernelGGL(rank_rows_block_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, dist, ld, n, m, row0, r1, r2, r1 != nullptr, rank, diag,
                       matrix_vec_ok(dist, ld) && (((uintptr_t)r2 & 15) == 0));
    return check_launch("snag_rank_rows_block");
}

extern "C" SNAG_API int snag_rank_cols_block(const float* dist, int64_t ld, int n, int m, const float* r1, const float* r2, const float* diag,
                                             int32_t* count, void* stream) {
    SNAG_REQUIRE(n >= 1 && m >= 1 && ld >= m, SNAG_ERR_ARG, "rank_cols_block: bad sizes");
    SNAG_REQUIRE(dist && diag && count, SNAG_ERR_ARG, "rank_cols_block: null pointer");
    SNAG_REQUIRE((r1 == nullptr) == (r2 == nullptr), SNAG_ERR_ARG, "rank_cols_block: r1 and r2 go together");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(count, 0, sizeof(int32_t) * m, st) != hipSuccess) return fail(SNAG_ERR_LAUNCH, "rank_cols_block: memset failed");
    const int chunks = std::min(kColChunks, n);
    const int rpc = ceil_div(n, chunks);
    if (matrix_vec_ok(dist, ld))
        hipLaunchKernelGGL(rank_cols4_kernel<true>, dim3(ceil_div(m, 256), ceil_div(n, rpc)), dim3(64), 0, st, dist, ld, n, m, r1, r2, r1 != nullptr, diag, rpc, count);
    else
        hipLaunchKernelGGL(rank_cols_block_kernel, dim3(ceil_div(m, 256), ceil_div(n, rpc)), dim3(256), 0, st, dist, ld, n, m, r1, r2, r1 != nullptr, diag, rpc,
                           count);
    return check_launch("snag_rank_cols_block");
}

extern "C" SNAG_API int snag_rank_summary(const int32_t* rank, int n, const int32_t* topk, int ntop, double* out, void* stream) {
    SNAG_REQUIRE(rank && topk && out && n >= 1 && ntop >= 0 && ntop <= 8, SNAG_ERR_ARG, "rank_summary: bad arguments");
    hipLaunchKernelGGL(rank_summary_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, rank, n, topk, ntop, out, (const int32_t*)nullptr);
    return check_launch("snag_rank_summary");
}

extern "C" SNAG_API int snag_rank_summary_pair(const int32_t* rank_a, const int32_t* rank_b, int n, const int32_t* topk, int ntop, double* out, void* stream) {
    SNAG_REQUIRE(rank_a && rank_b && topk && out && n >= 1 && ntop >= 0 && ntop <= 8, SNAG_ERR_ARG, "rank_summary_pair: bad arguments");
    hipLaunchKernelGGL(rank_summary_kernel, dim3(2), dim3(256), 0, (hipStream_t)stream, rank_a, n, topk, ntop, out, rank_b);
    return check_launch("snag_rank_summary_pair");
}

extern "C" SNAG_API int snag_argmin_rows_cols(const float* dist, int64_t ld, int n, int m, int32_t* row_argmin, int32_t* col_argmin, float* work,
                                              void* stream) {
    SNAG_REQUIRE(n >= 1 && m >= 1 && ld >= m && dist, SNAG_ERR_ARG, "argmin: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int vec = matrix_vec_ok(dist, ld);
    if (row_argmin) hipLaunchKernelGGL(argmin_rows_kernel, dim3(n), dim3(256), 0, st, dist, ld, n, m, row_argmin, vec);
    if (col_argmin) {
        SNAG_REQUIRE(work != nullptr, SNAG_ERR_ARG, "argmin: column arg-min needs the work buffer");
        const int chunks = std::min(kColChunks, n);
        const int rpc = ceil_div(n, chunks);
        const int nch = ceil_div(n, rpc);
        float* pv = work;
        int* pi = reinterpret_cast<int*>(work + (int64_t)kColChunks * m);
        if (vec) hipLaunchKernelGGL(argmin_cols_part4_kernel, dim3(ceil_div(m, 256), nch), dim3(64), 0, st, dist, ld, n, m, rpc, pv, pi);
        else hipLaunchKernelGGL(argmin_cols_part_kernel, dim3(ceil_div(m, 256), nch), dim3(256), 0, st, dist, ld, n, m, rpc, pv, pi);
        hipLaunchKernelGGL(argmin_cols_merge_kernel, dim3(ceil_div(m, 256)), dim3(256), 0, st, (const float*)pv, (const int*)pi, nch, m, col_argmin);
    }
    return check_launch("snag_argmin_rows_cols");
}
