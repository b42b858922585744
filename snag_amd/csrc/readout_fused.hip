// The read-out layer's forward as ONE launch (hidden_size 128, three tokens, the two projected tokens' second projections folded into
// their key / value weights): a workgroup owns 64 entities -- two halves of 32 rows, four waves each -- and a half walks the eight
// 32 x 128 x 128 products of the chain on its rows
//     q | k_g | v_g = x_g Wqkv^T ;  k_a | v_a = mid_a Wf_a^T ;  k_b | v_b = mid_b Wf_b^T ;  t1 = dropout(ctx Wo^T + bo) + x_g
// (reference: BertSelfAttention / BertSelfOutput, model/SNAG_tools.py:178-222; ProjectionHead's second Linear folded, layers.py:135-148)
// with the attention row of the read-out token and the LayerNorm in between, instead of three launches (grouped GEMM, attention,
// output projection + LayerNorm) whose prologues, loops and epilogues add up.
//   * The A operand of a product -- BM rows x 128 k of x_g, mid_a, mid_b, ctx -- is split ONCE into eight resident k-tile images
//     and serves one to three products; the next panel is already in registers when the current one retires.
//   * B (the weights) is split once per step into k-tile images in the order the launch consumes them (64 images of 12 KB, L2-resident;
//     written by the first workgroups of a split-K reduce launch of the step -- snag_fusion_fwd_prepare_reduce -- or by
//     readout_presplit_kernel); they stream through two LDS images as ONE 64-step pipeline across all eight products and both halves -- no
//     per-product prologue, no split arithmetic in the loop.
//   * Two waves per SIMD (eight waves per workgroup, 107 KB of LDS).  64 rows per four-wave workgroup = one wave per SIMD measured 84 us
//     against 82 for the three launches: every wait idles the matrix pipe.
//   * Accumulators go through a [BM][132] LDS tile; wave w then owns rows BM/4 w .. with column lane + 64 u in the lane --
//     the layout of attn_ro_fwd_kernel / ln_fwd_kernel, whose arithmetic (sum64 partial dot products, softmax, ln_row_forward) is
//     repeated statement by statement: q, k, v, probs, ctx, t1, hidden and stats are the three-launch path's, bit for bit.
//   * q stays in registers; a key product leaves only its scores (lane r of a register holds row r's); the softmax runs once, lane r
//     on row r; the value products then arrive in token order and ctx += p v grows in registers in attn_ro_fwd_kernel's association;
//     ctx goes back through the LDS tile into the A images of the output projection.
// Every activation the backward reads (qkv, probs, ctx, t1, stats) is written as before: the backward does not change.
#include "readout_fused.hip.h"

#include "fusion_kernels.hip.h"
#include "gemm.hip.h"

#ifndef SNAG_RO_BM
#define SNAG_RO_BM 32            // entities per workgroup (64: A/B)
#endif
#ifndef SNAG_RO_NH
#define SNAG_RO_NH 2             // 32-row halves per workgroup
#endif
#ifndef SNAG_RO_PRESPLIT
#define SNAG_RO_PRESPLIT 1       // 1: the weights are split into k-tile images by a launch of their own (readout_presplit_kernel) and copied into the LDS;
#endif                           // 0: the loop splits them itself -- one 16-byte load and two split_pair per thread and k-tile with 512 threads
                                 // (same box: 4.7 + 58.8 us against 63.9 us in one launch; step 487.5 against 489.8 us)
#ifndef SNAG_RO_PP
#define SNAG_RO_PP 1             // 1: the two halves of a workgroup run every k-loop half an iteration apart (ping-pong): a barrier cuts an iteration into its
#endif                           // memory half (fragment reads, the half's share of the next weight image, the stream's loads) and its matrix half (six MFMAs)
#ifndef SNAG_RO_STAMP
#define SNAG_RO_STAMP 0          // diagnostic build: cycle stamps of one wave printed per launch
#endif
#ifndef SNAG_RO_ABLATE
#define SNAG_RO_ABLATE 0         // timing experiments (wrong results): 1 no epilogue bodies, 2 no MFMAs, 4 no weight staging, 8 no fragment reads, 16 no output-projection tail
#endif

namespace snag {

namespace {

constexpr int kP = 132;                     // pitch of the LDS row tile (floats)
constexpr int kBImg = 6 * 128 * 16;         // one k-tile image of a 128-row operand (three bf16 planes, two k-halves)

using u4 = __attribute__((ext_vector_type(4))) uint32_t;
#if defined(__HIP_DEVICE_COMPILE__)
using lds_u2 = __attribute__((address_space(3))) uint2;
using lds_u4 = __attribute__((address_space(3))) u4;
#endif

// A buffer descriptor whose words are pinned to SGPRs: left to itself the compiler keeps a descriptor that went through a select or a
// spill in VGPRs and wraps every access in a waterfall loop (readfirstlane / compare / saveexec) -- 61 of them around the row stores of the
// first version of this kernel, ~400 cycles per store.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* base, uint32_t num_records) {
    const uint64_t b = reinterpret_cast<uint64_t>(base);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, (int)__builtin_amdgcn_readfirstlane(num_records), 0x00020000);
}

// a [BM, 128] K-contiguous panel in flight.  BM = 64: thread = (row t / 4, chunk t % 4) of every k-tile; BM = 32: the same for the
// threads' half (t / 128) of the k-tiles -- SplitStager's K-contiguous mapping.
template <int BM>
struct Panel {
    static constexpr int NL = BM / 8;                         // 16-byte loads per thread
    static constexpr int kImg = 6 * BM * 16;                  // one k-tile image
    u4 v[NL];
    static __device__ __forceinline__ int row_of(int tid) { return (tid & (4 * BM - 1)) >> 2; }
    static __device__ __forceinline__ int kt_of(int tid, int q) { return BM == 64 ? q : (tid >> 7) + 2 * q; }
    static __device__ __forceinline__ int wr_of(int tid) {
        const int c = tid & 3;
        return ((c & 1) * BM + row_of(tid)) * 16 + (c >> 1) * 8;
    }
    __device__ __forceinline__ void load(const float* __restrict__ P, int64_t ld, int m0, int N, int tid) {
#if defined(__HIP_DEVICE_COMPILE__)
        const char* ub = reinterpret_cast<const char*>(P + (int64_t)m0 * ld);
        const uint32_t voff = (uint32_t)(min(m0 + row_of(tid), N - 1) - m0) * (uint32_t)(ld * 4) + (uint32_t)(tid & 3) * 16u;     // (rows past the edge: the last row; never stored)
        const auto rsrc = uniform_rsrc(ub, 0xffffffffu);
#pragma unroll
        for (int q = 0; q < NL; ++q) v[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + (uint32_t)kt_of(tid, q) * 64u, 0, 0);
#endif
    }
    static __device__ __forceinline__ void chunk_store(unsigned char* aimg, int kt, int wr, float f0, float f1, float f2, float f3) {
#if defined(__HIP_DEVICE_COMPILE__)
        uint32_t h0, m0, l0, h1, m1, l1;
        split_pair(f0, f1, h0, m0, l0);
        split_pair(f2, f3, h1, m1, l1);
        auto* im = (__attribute__((address_space(3))) unsigned char*)aimg + kt * kImg + wr;
        *(lds_u2*)(im) = make_uint2(h0, h1);
        *(lds_u2*)(im + 2 * BM * 16) = make_uint2(m0, m1);
        *(lds_u2*)(im + 4 * BM * 16) = make_uint2(l0, l1);
#endif
    }
    __device__ __forceinline__ void store(unsigned char* aimg, int tid) const {
        const int wr = wr_of(tid);
#pragma unroll
        for (int q = 0; q < NL; ++q)
            chunk_store(aimg, kt_of(tid, q), wr, __uint_as_float(v[q][0]), __uint_as_float(v[q][1]), __uint_as_float(v[q][2]), __uint_as_float(v[q][3]));
    }
    // the same from the LDS row tile (ctx)
    static __device__ __forceinline__ void from_tile(const lds_float* T, unsigned char* aimg, int tid) {
        const int wr = wr_of(tid), row = row_of(tid), c = tid & 3;
#pragma unroll
        for (int q = 0; q < NL; ++q) {
            const int kt = kt_of(tid, q);
            const float4 f = *reinterpret_cast<const lds_float4*>(T + row * kP + kt * 16 + c * 4);
            chunk_store(aimg, kt, wr, f.x, f.y, f.z, f.w);
        }
    }
};

}  // namespace

// The weights of the eight products as k-tile images in consumption order (ro_weight_image_body, gemm.hip.h): a launch of its own when
// nobody prepared them (snag_fusion_fwd_prepare_reduce lets them ride in a split-K reduce launch of the step instead).
__global__ __launch_bounds__(256) void readout_presplit_kernel(RoWeightImages w) { ro_weight_image_body(w, (int)blockIdx.x); }

// NH: 32-row halves per workgroup (BM = 32 only).  NH = 2: eight waves, each half with its own A images and row tile, ONE weight stream
// for both -- the images cross from the L2 once per 64 entities instead of once per 32 (469 workgroups x 768 KB = 360 MB ran at the
// L2's pace: ~20 us of the launch), and a SIMD still holds two waves.
template <int BM, int NH>
__global__ __launch_bounds__(256 * NH, (BM == 32 && NH == 1) ? 2 : 1) void readout_fwd_fused_kernel(RoFusedArgs a) {
    static_assert(NH == 1 || BM == 32, "halves: 32 rows each");
    constexpr int M = 3, D = 128;
    constexpr int TN = BM / 32;                 // 32-column fragments per wave: BM = 64: 2 x 2 waves of 32 x 64; BM = 32: 1 x 4 waves of 32 x 32
    constexpr int RPW = BM / 4;                 // rows of the row tile a wave owns
    constexpr int kAImg = Panel<BM>::kImg;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kBImg + NH * (8 * kAImg + BM * kP * 4)];
    const int gtid = threadIdx.x, tid = gtid & 255, lane = tid & 63, wave = wave_id_uniform() & 3, half = wave_id_uniform() >> 2;
    unsigned char* const bbuf = lds;
    unsigned char* const aimg = lds + 2 * kBImg + half * 8 * kAImg;
    lds_float* const T = (lds_float*)(lds + 2 * kBImg + NH * 8 * kAImg + half * BM * kP * 4);
    const int li = lane & 31, lh = lane >> 5;
    const int wm0 = BM == 64 ? (wave >> 1) * 32 : 0, wn0 = BM == 64 ? (wave & 1) * 64 : wave * 32;
    const int N = a.N, H = a.H, g = a.g, d = D / H, HMM = H * M * M;
    const int m_first = (blockIdx.x * NH + half) * BM;          // (the last workgroup's second half may start behind the last entity:
    const int m0 = min(m_first, N - 1);                         //  it keeps the barriers and the weight stream company, loads the last row, stores nothing)
    const int rw = RPW * wave;                                 // the wave's rows of the row tile
    const int valid = max(0, min(BM, N - m_first));            // rows of this half that exist

    f32x16 acc[TN];
    [[maybe_unused]] int cur_tile = 0;                         // the product that is running
#if SNAG_RO_PRESPLIT
    constexpr int NB = NH == 1 ? 3 : 2;                        // 16-byte pieces of an image per thread (NH = 2: 768 pieces over 512 threads)
    u4 br[4][NB];                                              // four register sets of the weight stream
    // image i of the 64-step weight stream -> register set S; past the end: zeros through an empty descriptor, no traffic
    auto fetch_b = [&](auto S, int i) {
        constexpr int set = decltype(S)::value;
        const int in = __builtin_amdgcn_readfirstlane(i < 8 * kRoTiles ? i : 0);
        const auto rsrc = uniform_rsrc(a.wimg + (size_t)in * kBImg, i < 8 * kRoTiles ? kBImg : 0);
#pragma unroll
        for (int q = 0; q < NB; ++q) br[set][q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (uint32_t)(gtid + 256 * NH * q) * 16u, 0, 0);      // (pieces >= 768: out of range, zeros)
    };
    auto stage_b = [&](auto S, unsigned char* image) {
        constexpr int set = decltype(S)::value;
#if defined(__HIP_DEVICE_COMPILE__)
        auto* im = (__attribute__((address_space(3))) unsigned char*)image + gtid * 16;
#pragma unroll
        for (int q = 0; q < NB; ++q)
            if (NH == 1 || gtid + 512 * q < 768) *(lds_u4*)(im + 4096 * NH * q) = br[set][q];
#endif
    };
#else
    // The weight stream in f32: thread = (weight row t / 4 [+ 64 with 256 threads], chunk t % 4) of a k-tile -- SplitStager<128, K-contiguous>'s
    // mapping and split_pair, i.e. the bits the GEMM kernels put into their LDS images.  wcur / wnext: the weight rows of this product and
    // the next one (set once per product: no scalar load, hence no lgkmcnt wait, inside the loop).
    constexpr int NB = 2 / NH;
    u4 br[4][NB];
    const char *wcur = reinterpret_cast<const char*>(a.t[0].W), *wnext = wcur;
    const uint32_t bvoff = (uint32_t)(gtid >> 2) * (uint32_t)(a.ldw * 4) + (uint32_t)(gtid & 3) * 16u;
    const int bwr = (((gtid & 3) & 1) * 128 + (gtid >> 2)) * 16 + ((gtid & 3) >> 1) * 8;
    // k-tile kt (0 .. 12) of the product that is running: past its eighth k-tile the next product's; past the last product: zeros, no traffic
    auto fetch_b = [&](auto S, int i) {
        constexpr int set = decltype(S)::value;
        const int kt = i & 7;
        const bool next = (i >> 3) != cur_tile;
        const auto rsrc = uniform_rsrc(next ? wnext : wcur, i < 8 * kRoTiles ? 0xffffffffu : 0u);
#pragma unroll
        for (int q = 0; q < NB; ++q) br[set][q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, bvoff + (uint32_t)(64 * q) * (uint32_t)(a.ldw * 4), (uint32_t)kt * 64u, 0);
    };
    auto stage_b = [&](auto S, unsigned char* image) {
        constexpr int set = decltype(S)::value;
#if defined(__HIP_DEVICE_COMPILE__)
        auto* im = (__attribute__((address_space(3))) unsigned char*)image + bwr;
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            uint32_t h0, m0_, l0, h1, m1, l1;
            split_pair(__uint_as_float(br[set][q][0]), __uint_as_float(br[set][q][1]), h0, m0_, l0);
            split_pair(__uint_as_float(br[set][q][2]), __uint_as_float(br[set][q][3]), h1, m1, l1);
            *(lds_u2*)(im + q * 64 * 16) = make_uint2(h0, h1);
            *(lds_u2*)(im + q * 64 * 16 + 2 * 128 * 16) = make_uint2(m0_, m1);
            *(lds_u2*)(im + q * 64 * 16 + 4 * 128 * 16) = make_uint2(l0, l1);
        }
#endif
    };
#endif
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    using S3 = std::integral_constant<int, 3>;
    // rows of an [N, *] output through a descriptor that ends behind the last existing row: the stores need no branch (their number is
    // static, so the waits on the weight stream's loads stay counted)
    auto rows_rsrc = [&](float* base, int64_t ld) {
        return uniform_rsrc(base + (int64_t)m0 * ld, (uint32_t)((int64_t)valid * ld * 4));
    };
    // One k-tile of the stream (image i = i0 + kt): fragments of image i out of LDS image i & 1, image i + 1 (register set S = (i + 1) & 3)
    // into the other one, image i + 5 requested into set S (four sets: a request has four k-tiles to land), the six products in
    // mfma_split's order, barrier.  (Requesting the NEXT k-tile's fragments in front of the products -- register double-buffered, the
    // products pinned in front of the barrier -- measured slower on the same box: 67.0 vs 60.6 us.)
    auto sub_iteration = [&](int kt, auto S, int i0) {
        Split3 af, bf[TN];
        const int cur = kt & 1;
#if !(SNAG_RO_ABLATE & 8)
        load_split_fragment<BM>(aimg + kt * kAImg, wm0 + li, lh, af);
#pragma unroll
        for (int j = 0; j < TN; ++j) load_split_fragment<128>(bbuf + cur * kBImg, wn0 + 32 * j + li, lh, bf[j]);
#endif
#if !(SNAG_RO_ABLATE & 4)
        stage_b(S, bbuf + (cur ^ 1) * kBImg);
        fetch_b(S, i0 + kt + 5);
#endif
#if defined(__HIP_DEVICE_COMPILE__) && SNAG_RO_PP
        // PING-PONG (NH = 2): everything above is the MEMORY half of the iteration -- it ends with the fragments in registers and this half's
        // share of image i + 1 in the LDS -- everything below the MATRIX half.  The second half-workgroup entered the k-loop one barrier late
        // (see the loop), so while one half multiplies the other one reads and stages: a SIMD's matrix pipe always has exactly one of its two
        // waves on it, and the LDS traffic of an iteration (48 KB of fragment reads, 12 KB of image writes) runs in the MFMAs' shadow instead
        // of in front of them.  Image i + 1 is complete one barrier after the late half's memory half, one barrier before the early half reads it;
        // the image it replaces was last read two barriers earlier.
        if constexpr (NH == 2) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !(SNAG_RO_ABLATE & 2)
#define SNAG_RO_TERM(X, Y) _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af.X, bf[j].Y, acc[j], 0, 0, 0);
        SNAG_RO_TERM(l, h)
        SNAG_RO_TERM(h, l)
        SNAG_RO_TERM(m, m)
        SNAG_RO_TERM(m, h)
        SNAG_RO_TERM(h, m)
        SNAG_RO_TERM(h, h)
#undef SNAG_RO_TERM
#endif
#if defined(__HIP_DEVICE_COMPILE__) && SNAG_RO_PP
        if constexpr (NH == 2) {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        } else {
            __syncthreads();
        }
#else
        __syncthreads();
#endif
    };

    Panel<BM> pr;
    pr.load(a.panel[a.t[0].panel], a.ldp[a.t[0].panel], m0, N, tid);
    fetch_b(S0{}, 0);
    fetch_b(S1{}, 1);
    fetch_b(S2{}, 2);
    fetch_b(S3{}, 3);
    pr.store(aimg, tid);
    stage_b(S0{}, bbuf);
    fetch_b(S0{}, 4);
    __syncthreads();

    float qr[RPW][2], o[RPW][2], sc[M][kRoMaxHeads];     // q rows ; ctx rows ; scores, then probabilities: lane r of sc[j][h] = row rw + r
#pragma unroll
    for (int j = 0; j < M; ++j)
#pragma unroll
        for (int h = 0; h < kRoMaxHeads; ++h) sc[j][h] = 0.f;
    int hsel[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) hsel[u] = (lane + 64 * u) / d;
    int n_keys = 0, n_vals = 0;
    // the attention dropout's keep flags of row rw + lane (lanes < RPW), requested now: the softmax finds them in registers
    int kp[kRoMaxHeads][M];
    {
        const int64_t n = m0 + rw + lane;
        const bool mine = lane < RPW && rw + lane < valid;
#pragma unroll
        for (int h = 0; h < kRoMaxHeads; ++h)
#pragma unroll
            for (int j = 0; j < M; ++j) kp[h][j] = (a.keep_probs != nullptr && mine && h < H) ? (int)a.keep_probs[n * HMM + h * M * M + g * M + j] : 1;
    }
#if SNAG_RO_STAMP
    __shared__ unsigned long long stamp[kRoTiles][5];        // (in LDS: a register array indexed by the running product would unroll the loop)
    const unsigned long long t_begin = __builtin_readcyclecounter();
#define SNAG_RO_ST(I) do { __builtin_amdgcn_sched_barrier(0); if (gtid == 0) stamp[tile][I] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SNAG_RO_ST(I) do { } while (0)
#endif

    for (int tile = 0; tile < kRoTiles; ++tile) {
        const RoTile& t = a.t[tile];
        SNAG_RO_ST(0);
        // the next product's A panel, if it is another one, and this product's bias travel while the product runs
#if !SNAG_RO_PRESPLIT
        cur_tile = tile;
        wcur = reinterpret_cast<const char*>(t.W);
        wnext = reinterpret_cast<const char*>(a.t[min(tile + 1, kRoTiles - 1)].W);
#endif
        const int next_panel = tile + 1 < kRoTiles ? a.t[tile + 1].panel : t.panel;
        const bool switch_panel = next_panel != t.panel;
        if (switch_panel && next_panel < 3) pr.load(a.panel[next_panel], a.ldp[next_panel], m0, N, tid);
        const bool hb = t.bias != nullptr;
        float bv[2] = {0.f, 0.f};
        if (hb) {
            bv[0] = t.bias[lane];
            bv[1] = t.bias[lane + 64];
        }
        // (the output projection's side inputs -- keep mask and residual rows, cold -- of the wave's first eight rows as well: same box,
        // 60.6 vs 65.8 us)
        const bool hm = a.keep_out != nullptr, ha = a.resid != nullptr;
        float av0[8][2];
        uint8_t kv0[8][2];
        if (t.kind == 3) {
            if (hm) {
#pragma unroll
                for (int q = 0; q < 8; ++q)
#pragma unroll
                    for (int u = 0; u < 2; ++u) kv0[q][u] = a.keep_out[(int64_t)min(m0 + rw + q, N - 1) * a.ldkeep + lane + 64 * u];
            }
            if (ha) {
#pragma unroll
                for (int q = 0; q < 8; ++q)
#pragma unroll
                    for (int u = 0; u < 2; ++u) av0[q][u] = a.resid[(int64_t)min(m0 + rw + q, N - 1) * a.ldres + lane + 64 * u];
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        const int i0 = tile * 8;
#if defined(__HIP_DEVICE_COMPILE__) && SNAG_RO_PP
        if (NH == 2 && half == 1) __builtin_amdgcn_s_barrier();          // the late half: its memory halves face the early half's matrix halves
#endif
#pragma unroll 1
        for (int kt = 0; kt < 8; kt += 4) {
            sub_iteration(kt, S1{}, i0);
            sub_iteration(kt + 1, S2{}, i0);
            sub_iteration(kt + 2, S3{}, i0);
            sub_iteration(kt + 3, S0{}, i0);
        }
#if defined(__HIP_DEVICE_COMPILE__) && SNAG_RO_PP
        if (NH == 2 && half == 0) __builtin_amdgcn_s_barrier();          // the halves meet again: the epilogues below run side by side
#endif
        SNAG_RO_ST(1);
        // accumulators -> the row tile
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) T[(wm0 + (e & 3) + 8 * (e >> 2) + 4 * lh) * kP + wn0 + 32 * j + li] = acc[j][e];
        __syncthreads();
        SNAG_RO_ST(2);
        const auto orsrc = rows_rsrc(t.out, t.ldo);
        const uint32_t ovoff = (uint32_t)rw * (uint32_t)(t.ldo * 4) + (uint32_t)lane * 4u;
        const uint32_t ostep = (uint32_t)(t.ldo * 4);
        if (SNAG_RO_ABLATE & 1) {
        } else if (t.kind == 0) {
            // ---- query rows: bias, store, keep ------------------------------------------------------------------------------------
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float v = T[(rw + r) * kP + lane + 64 * u];
                    qr[r][u] = hb ? v + bv[u] : v;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(qr[r][u]), orsrc, ovoff + r * ostep, 256u * u, 0);
                }
            }
        } else if (t.kind == 1) {
            // ---- key rows: bias, store, the scores of the wave's rows --------------------------------------------------------------
            // attn_ro_fwd_kernel forms a score as sum64 of the lanes' partial dot products.  sum64 is, bit for bit, the butterfly over the
            // lane bits 8, 4, 2, 1, 16, 32 (every level adds a lane's value and its partner's: f32 addition commutes), so the RPW x 4
            // partials of a key product are summed by ONE reduce-scatter in that bit order -- 31 exchanges, halving the registers at each
            // level, instead of 32 chains of six dependent DPP adds and a readlane: lane L ends with the total of
            // (row (L >> 1) & 7, head 2 (L & 1) + ((L >> 4) & 1))  [16 rows per wave: (row L & 15, head 2 ((L >> 4) & 1) + (L >> 5))].
            const int tok = t.token;
            float pv[RPW * kRoMaxHeads];
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                float x[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float v = T[(rw + r) * kP + lane + 64 * u];
                    x[u] = hb ? v + bv[u] : v;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x[u]), orsrc, ovoff + r * ostep, 256u * u, 0);
                }
#pragma unroll
                for (int h = 0; h < kRoMaxHeads; ++h) {
                    const int lo = h * d, hi = lo + d;         // (h >= H: an empty column range, the partial is zero)
                    float part = 0.f;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int c = lane + 64 * u;
                        part += (h < H && c >= lo && c < hi) ? qr[r][u] * x[u] : 0.f;
                    }
                    pv[r * kRoMaxHeads + h] = part;
                }
            }
            static_assert(RPW * kRoMaxHeads == 32 || RPW * kRoMaxHeads == 64, "reduce-scatter over five or six lane bits");
            {
                constexpr int kBits[6] = {8, 4, 2, 1, 16, 32};
                int n = RPW * kRoMaxHeads;
#pragma unroll
                for (int sidx = 0; sidx < 6; ++sidx) {
                    const int b = kBits[sidx];
                    const bool up = (lane & b) != 0;
                    if (n > 1) {
#pragma unroll
                        for (int i = 0; i < RPW * kRoMaxHeads / 2; ++i) {
                            if (i < n / 2) {
                                const float lo_v = pv[i], hi_v = pv[i + n / 2];
                                const float got = __shfl_xor(up ? lo_v : hi_v, b);
                                pv[i] = (up ? hi_v : lo_v) + got;
                            }
                        }
                        n /= 2;
                    } else {
                        pv[0] += __shfl_xor(pv[0], b);
                    }
                }
            }
#pragma unroll
            for (int h = 0; h < kRoMaxHeads; ++h) {
                if (h < H) {
                    // row r's total of head h -> lane r
                    const int src = RPW == 8 ? 2 * (lane & 7) + (h >> 1) + 16 * (h & 1) : (lane & 15) + 16 * (h >> 1) + 32 * (h & 1);
                    const float s = __shfl(pv[0], src) * a.inv_sqrt_d;
#pragma unroll
                    for (int j = 0; j < M; ++j)
                        if (tok == j) sc[j][h] = s;
                }
            }
            if (++n_keys == M) {
                // every score is here: softmax, dropout and probs with lane r working on row rw + r (the same operations on the same values
                // as head_scores_row + the keep step of attn_ro_fwd_kernel; sc turns into the probabilities)
                const int64_t n = m0 + rw + lane;
                const bool mine = lane < RPW && rw + lane < valid;
#pragma unroll
                for (int h = 0; h < kRoMaxHeads; ++h) {
                    if (h < H) {
                        float p[M];
#pragma unroll
                        for (int j = 0; j < M; ++j) p[j] = sc[j][h];
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
                        if (a.keep_probs != nullptr) {
#pragma unroll
                            for (int j = 0; j < M; ++j) {
                                p[j] = kp[h][j] ? p[j] * a.kscale : 0.f;
                            }
                        }
#pragma unroll
                        for (int j = 0; j < M; ++j) {
                            if (mine) a.probs[(n * H + h) * (int64_t)(M * M) + g * M + j] = p[j];
                            sc[j][h] = p[j];
                        }
                    }
                }
            }
        } else if (t.kind == 2) {
            // ---- value rows (token order): bias, store, ctx += p v -- the association of attn_ro_fwd_kernel's sum over the tokens ------
            const int tok = t.token;
            const bool first = n_vals == 0;
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                float ph[kRoMaxHeads];
#pragma unroll
                for (int h = 0; h < kRoMaxHeads; ++h) {
                    float s = sc[0][h];
#pragma unroll
                    for (int j = 1; j < M; ++j) s = tok == j ? sc[j][h] : s;
                    ph[h] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), r));
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float v = T[(rw + r) * kP + lane + 64 * u];
                    const float x = hb ? v + bv[u] : v;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x), orsrc, ovoff + r * ostep, 256u * u, 0);
                    float pv = ph[0];
#pragma unroll
                    for (int h = 1; h < kRoMaxHeads; ++h) pv = hsel[u] == h ? ph[h] : pv;
                    o[r][u] = (first ? 0.f : o[r][u]) + pv * x;
                }
            }
            if (++n_vals == M) {
                const auto crsrc = rows_rsrc(a.ctx, D);
#pragma unroll
                for (int r = 0; r < RPW; ++r)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o[r][u]), crsrc, (uint32_t)(rw + r) * (D * 4) + (uint32_t)lane * 4u, 256u * u, 0);
                        T[(rw + r) * kP + lane + 64 * u] = o[r][u];
                    }
            }
        } else if (!(SNAG_RO_ABLATE & 16)) {
            // ---- output projection: Epilogue::one's order (bias, keep mask, residual), the pre-LayerNorm row, ln_row_forward ----------
            // (the tail of gemm_tile_ln_kernel: eight rows' side inputs requested together)
            for (int r0 = 0; r0 < RPW; r0 += 8) {
                float x[8][2], av[8][2];
                uint8_t kv[8][2];
                if (r0 == 0) {
#pragma unroll
                    for (int q = 0; q < 8; ++q)
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            kv[q][u] = kv0[q][u];
                            av[q][u] = av0[q][u];
                        }
                } else {
                if (hm) {
#pragma unroll
                    for (int q = 0; q < 8; ++q)
#pragma unroll
                        for (int u = 0; u < 2; ++u) kv[q][u] = a.keep_out[(int64_t)min(m0 + rw + r0 + q, N - 1) * a.ldkeep + lane + 64 * u];
                }
                if (ha) {
#pragma unroll
                    for (int q = 0; q < 8; ++q)
#pragma unroll
                        for (int u = 0; u < 2; ++u) av[q][u] = a.resid[(int64_t)min(m0 + rw + r0 + q, N - 1) * a.ldres + lane + 64 * u];
                }
                }
#pragma unroll
                for (int q = 0; q < 8; ++q)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        float v = T[(rw + r0 + q) * kP + lane + 64 * u];
                        if (hb) v += bv[u];
                        if (hm) v = kv[q][u] ? v * a.kscale : 0.f;
                        if (ha) v += av[q][u];
                        x[q][u] = v;
                    }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int m = m0 + rw + r0 + q;
                    if (rw + r0 + q < valid) {
                        float y[2], mean, rstd;
                        ln_row_forward<2>(x[q], D, lane, a.ln_g, a.ln_b, a.eps, y, mean, rstd);
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            t.out[(int64_t)m * t.ldo + lane + 64 * u] = x[q][u];
                            a.hidden[(int64_t)m * D + lane + 64 * u] = y[u];
                        }
                        if (lane == 0) {
                            a.stats[2 * (int64_t)m] = mean;
                            a.stats[2 * (int64_t)m + 1] = rstd;
                        }
                    }
                }
            }
        }
        SNAG_RO_ST(3);
        // ---- the next product's A panel (every fragment read of the old one is behind a barrier) ---------------------------------------
        if (switch_panel) {
            if (next_panel < 3) {
                pr.store(aimg, tid);
            } else {                                  // ctx, from the row tile
                __syncthreads();
                Panel<BM>::from_tile(T, aimg, tid);
            }
            __syncthreads();
        }
        SNAG_RO_ST(4);
    }
#if SNAG_RO_STAMP
    if (blockIdx.x == 100 && gtid == 0) {
        printf("ro stamps (cycles since kernel start: tile begin | k-loop end | tile in LDS | epilogue end | panel end)\n");
        for (int i = 0; i < kRoTiles; ++i)
            printf("  tile %d kind %d: %6llu %6llu %6llu %6llu %6llu\n", i, a.t[i].kind, stamp[i][0] - t_begin, stamp[i][1] - t_begin, stamp[i][2] - t_begin, stamp[i][3] - t_begin, stamp[i][4] - t_begin);
    }
#endif
}

// ---- backward: d ctx + attention backward -----------------------------------------------------------------------------------------
// Wo^T as k-tile images: image kt holds B(n = input column i, k = output row o) = Wo[o * ldw + i] for o = 16 kt .. 16 kt + 15 --
// SplitStager<128, row-contiguous>'s k permutation inside the k-tile is the K-contiguous one's (chunk c = four consecutive k), so the
// images are the ones gemm_tile_kernel<.., B row-contiguous> builds in its LDS.
__global__ __launch_bounds__(256) void readout_presplit_t_kernel(RoBwdArgs a) {
    const int kt = blockIdx.x, tid = threadIdx.x;
    const int row = tid >> 2, c = tid & 3;
    unsigned char* im = a.wimg + (size_t)kt * kBImg + ((c & 1) * 128 + row) * 16 + (c >> 1) * 8;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float f[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) f[j] = a.wo[(int64_t)(kt * 16 + 4 * c + j) * a.ldw + row + 64 * i];
        uint32_t h0, m0, l0, h1, m1, l1;
        split_pair(f[0], f[1], h0, m0, l0);
        split_pair(f[2], f[3], h1, m1, l1);
        *reinterpret_cast<uint2*>(im + i * 64 * 16) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(im + i * 64 * 16 + 2 * 128 * 16) = make_uint2(m0, m1);
        *reinterpret_cast<uint2*>(im + i * 64 * 16 + 4 * 128 * 16) = make_uint2(l0, l1);
    }
}

template <int NH>
__global__ __launch_bounds__(256 * NH) void readout_bwd_attn_kernel(RoBwdArgs a) {
    constexpr int M = 3, D = 128, BM = 32, RPW = 8;
    constexpr int kAImg = Panel<BM>::kImg;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kBImg + NH * (8 * kAImg + BM * kP * 4)];
    __shared__ float red[NH][4][3][D];
    const int gtid = threadIdx.x, tid = gtid & 255, lane = tid & 63, wave = wave_id_uniform() & 3, half = wave_id_uniform() >> 2;
    unsigned char* const bbuf = lds;
    unsigned char* const aimg = lds + 2 * kBImg + half * 8 * kAImg;
    lds_float* const T = (lds_float*)(lds + 2 * kBImg + NH * 8 * kAImg + half * BM * kP * 4);
    const int li = lane & 31, lh = lane >> 5, wn0 = wave * 32;
    const int N = a.N;
    const int m_first = (blockIdx.x * NH + half) * BM, m0 = min(m_first, N - 1);
    const int rw = RPW * wave, valid = max(0, min(BM, N - m_first));
    constexpr int NB = NH == 1 ? 3 : 2;
    u4 br[4][NB];
    auto fetch_b = [&](auto S, int i) {
        constexpr int set = decltype(S)::value;
        const auto rsrc = uniform_rsrc(a.wimg + (size_t)(i < 8 ? i : 0) * kBImg, i < 8 ? kBImg : 0);
#pragma unroll
        for (int q = 0; q < NB; ++q) br[set][q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (uint32_t)(gtid + 256 * NH * q) * 16u, 0, 0);
    };
    auto stage_b = [&](auto S, unsigned char* image) {
        constexpr int set = decltype(S)::value;
#if defined(__HIP_DEVICE_COMPILE__)
        auto* im = (__attribute__((address_space(3))) unsigned char*)image + gtid * 16;
#pragma unroll
        for (int q = 0; q < NB; ++q)
            if (NH == 1 || gtid + 512 * q < 768) *(lds_u4*)(im + 4096 * NH * q) = br[set][q];
#endif
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    using S3 = std::integral_constant<int, 3>;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    auto sub_iteration = [&](int kt, auto S) {
        Split3 af, bf;
        const int cur = kt & 1;
        load_split_fragment<BM>(aimg + kt * kAImg, li, lh, af);
        load_split_fragment<128>(bbuf + cur * kBImg, wn0 + li, lh, bf);
        stage_b(S, bbuf + (cur ^ 1) * kBImg);
        fetch_b(S, kt + 5);
#if defined(__HIP_DEVICE_COMPILE__)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af.l, bf.h, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af.h, bf.l, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af.m, bf.m, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af.m, bf.h, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af.h, bf.m, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af.h, bf.h, acc, 0, 0, 0);
#endif
        __syncthreads();
    };
    // (the first two entities' rows travel during the product: they do not depend on d ctx)
    const int n_mine = max(0, min(RPW, valid - rw));
    const int64_t n0 = (int64_t)m_first + rw, n_last = (int64_t)N - 1;
    AttnRoRows<M, 2> nxt[2] = {attn_ro_load<M, 2>(a.qkv, min(n0, n_last), D, a.H, a.g, a.keep_probs, lane),
                               attn_ro_load<M, 2>(a.qkv, min(n0 + 1, n_last), D, a.H, a.g, a.keep_probs, lane)};
    Panel<BM> pr;
    pr.load(a.dz, a.lddz, m0, N, tid);
    fetch_b(S0{}, 0);
    fetch_b(S1{}, 1);
    fetch_b(S2{}, 2);
    fetch_b(S3{}, 3);
    pr.store(aimg, tid);
    stage_b(S0{}, bbuf);
    fetch_b(S0{}, 4);
    __syncthreads();
#pragma unroll 1
    for (int kt = 0; kt < 8; kt += 4) {
        sub_iteration(kt, S1{});
        sub_iteration(kt + 1, S2{});
        sub_iteration(kt + 2, S3{});
        sub_iteration(kt + 3, S0{});
    }
    // d ctx rows -> the row tile
#pragma unroll
    for (int e = 0; e < 16; ++e) T[((e & 3) + 8 * (e >> 2) + 4 * lh) * kP + wn0 + li] = acc[e];
    __syncthreads();
    // the wave's eight entities, one after the other (attn_ro_bwd_kernel's arithmetic; d ctx from the row tile)
    Cols<2> sq, sk, sv;
#pragma unroll
    for (int u = 0; u < 2; ++u) sq.v[u] = sk.v[u] = sv.v[u] = 0.f;
    // Two entities at a time (their sum64 chains interleave), the next pair's rows requested before the current pair's arithmetic: eight waves
    // per CU do not hide a round trip or a dependent chain by themselves.
#pragma unroll 1
    for (int r = 0; r < n_mine; r += 2) {
        const AttnRoRows<M, 2> cur[2] = {nxt[0], nxt[1]};
        nxt[0] = attn_ro_load<M, 2>(a.qkv, min(n0 + r + 2, n_last), D, a.H, a.g, a.keep_probs, lane);
        nxt[1] = attn_ro_load<M, 2>(a.qkv, min(n0 + r + 3, n_last), D, a.H, a.g, a.keep_probs, lane);
        Cols<2> go[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int rr = min(rw + r + e, BM - 1);
            go[e].v[0] = T[rr * kP + lane];
            go[e].v[1] = T[rr * kP + lane + 64];
        }
        const int64_t ns[2] = {n0 + r, n0 + r + 1};
        const bool live[2] = {true, r + 1 < n_mine};
        attn_ro_bwd_entities<M, 2, 2>(cur, go, ns, live, D, a.H, a.g, a.inv_sqrt_d, a.keep_probs, a.kscale, a.dq_g, a.dkv, a.ro3, lane, sq, sk, sv);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        red[half][wave][0][lane + 64 * u] = sq.v[u];
        red[half][wave][1][lane + 64 * u] = sk.v[u];
        red[half][wave][2][lane + 64 * u] = sv.v[u];
    }
    __syncthreads();
    if (m_first < N)
        for (int i = tid; i < 3 * D; i += 256) {
            const int which = i / D, c = i % D;
            a.bias_part[(int64_t)(blockIdx.x * NH + half) * 3 * D + which * D + c] =
                ((red[half][0][which][c] + red[half][1][which][c]) + red[half][2][which][c]) + red[half][3][which][c];
        }
}

int readout_bwd_attn_fused(const RoBwdArgs& a, hipStream_t st) {
    if (a.N <= 0) return SNAG_OK;
    if (!a.images_ready) hipLaunchKernelGGL(readout_presplit_t_kernel, dim3(8), dim3(256), 0, st, a);
    hipLaunchKernelGGL((readout_bwd_attn_kernel<SNAG_RO_NH>), dim3((unsigned)ceil_div(a.N, 32 * SNAG_RO_NH)), dim3(256 * SNAG_RO_NH), 0, st, a);
    return check_launch("fusion attention backward (read-out, with its d ctx product)");
}

int64_t readout_fused_image_bytes() { return (int64_t)(8 * kRoTiles + 8) * kBImg; }      // + Wo^T for the backward

RoWeightImages readout_weight_images(const RoFusedArgs& a) {
    RoWeightImages w{};
    for (int i = 0; i < kRoTiles; ++i) w.W[i] = a.t[i].W;
    w.Wt = nullptr;
    w.ldw = a.ldw;
    w.n_images = 8 * kRoTiles;
    w.wimg = a.wimg;
    return w;
}

int readout_fwd_fused(const RoFusedArgs& a, hipStream_t st, bool images_ready) {
    if (a.N <= 0) return SNAG_OK;
#if SNAG_RO_PRESPLIT
    if (!images_ready) hipLaunchKernelGGL(readout_presplit_kernel, dim3(8 * kRoTiles), dim3(256), 0, st, readout_weight_images(a));
#endif
    hipLaunchKernelGGL((readout_fwd_fused_kernel<SNAG_RO_BM, SNAG_RO_NH>), dim3((unsigned)ceil_div(a.N, SNAG_RO_BM * SNAG_RO_NH)), dim3(256 * SNAG_RO_NH), 0, st, a);
    return check_launch("fusion read-out forward (one launch)");
}

}  // namespace snag
