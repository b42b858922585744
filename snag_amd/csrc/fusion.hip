// snag_fusion_fwd / snag_fusion_bwd: host-side orchestration of the Mformer layer stack on one stream.
// reference: MformerFusion.forward model/SNAG_tools.py:32-51, BertLayer :279-298.
#include <algorithm>

#include "fusion_kernels.hip.h"
#include "readout_fused.hip.h"

// The one-launch read-out forward / the backward's fused launch are taken from this many entities on (a workgroup walks the whole chain: its
// time does not shrink with N).  One rank's share of an R-way row-sharded DB15K step, separate launches vs one launch (tools/ab_shard.sh):
// forward 53.4 / 48.9 us at 7,500 rows, 42.1 / 45.8 at 3,750, 38.2 / 44.9 at 1,875; backward 27.1 / 28.3, 20.7 / 25.0, 17.2 / 24.6 (36.3 / 32.5 at 15,000).
#ifndef SNAG_RO_FWD_MIN_ROWS
#define SNAG_RO_FWD_MIN_ROWS 5120
#endif
#ifndef SNAG_RO_BWD_MIN_ROWS
#define SNAG_RO_BWD_MIN_ROWS 10240
#endif
// ... and only when their workgroups (64 entities each, one per CU) fill the last round of the chip to 70 %: 20,000 entities = 313 workgroups
// = two rounds of which the second is a quarter full, i.e. twice the time of 16,384 entities.
static bool ro_rounds_filled(int N) {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
        else cus = 256;
    }
    const int wgs = snag::ceil_div(N, 64), rounds = snag::ceil_div(wgs, cus);
    return wgs * 10 >= rounds * cus * 7;
}

#ifndef SNAG_RO_BWD_FUSED
#define SNAG_RO_BWD_FUSED 1      // A/B: 0 keeps the d ctx GEMM and the attention backward as two launches in the read-out layer's backward
#endif

namespace snag {

// ---- activation layout inside `save` (identical in forward and backward) ----------------------
struct LayerSave {
    float *qkv, *probs, *ctx, *t1, *st1, *y1, *u, *g, *t2, *st2, *y2;
};

struct SaveLayout {
    int64_t per_layer;
    int64_t NM, D, I, PP;
    bool ffn;
    SaveLayout(int N, int M, int D_, int H, int I_, bool ffn_) : NM((int64_t)N * M), D(D_), I(I_), PP((int64_t)N * H * M * M), ffn(ffn_) {
        per_layer = NM * 3 * D + PP + NM * D /*ctx*/ + NM * D /*t1*/ + NM * 2 + NM * D /*y1*/;
        if (ffn) per_layer += 2 * NM * I + NM * D /*t2*/ + NM * 2 + NM * D /*y2*/;
        per_layer = (per_layer + 3) / 4 * 4;
    }
    LayerSave at(float* base, int l) const {
        LayerSave s{};
        float* p = base + per_layer * l;
        s.qkv = p; p += NM * 3 * D;
        s.ctx = p; p += NM * D;
        s.t1 = p; p += NM * D;
        s.y1 = p; p += NM * D;
        if (ffn) {
            s.u = p; p += NM * I;
            s.g = p; p += NM * I;
            s.t2 = p; p += NM * D;
            s.y2 = p; p += NM * D;
            s.st2 = p; p += NM * 2;
        }
        s.st1 = p; p += NM * 2;
        s.probs = p;
        return s;
    }
};

static int pick_nu(int D) { return D <= 128 ? 2 : (D <= 320 ? 5 : 8); }

// dispatch a kernel template<int M, int NU> over runtime (M, D)
#define SNAG_MNU_CASE(KERNEL, MM, NU, grid, st, ...) \
    hipLaunchKernelGGL((KERNEL<MM, NU>), grid, dim3(256), 0, st, __VA_ARGS__)
#define SNAG_MNU_M(KERNEL, NU, M, grid, st, ...)                          \
    switch (M) {                                                          \
        case 1: SNAG_MNU_CASE(KERNEL, 1, NU, grid, st, __VA_ARGS__); break; \
        case 2: SNAG_MNU_CASE(KERNEL, 2, NU, grid, st, __VA_ARGS__); break; \
        case 3: SNAG_MNU_CASE(KERNEL, 3, NU, grid, st, __VA_ARGS__); break; \
        case 4: SNAG_MNU_CASE(KERNEL, 4, NU, grid, st, __VA_ARGS__); break; \
        case 5: SNAG_MNU_CASE(KERNEL, 5, NU, grid, st, __VA_ARGS__); break; \
        default: SNAG_MNU_CASE(KERNEL, 6, NU, grid, st, __VA_ARGS__); break; \
    }
#define SNAG_MNU(KERNEL, M, D, grid, st, ...)                                  \
    switch (pick_nu(D)) {                                                      \
        case 2: SNAG_MNU_M(KERNEL, 2, M, grid, st, __VA_ARGS__); break;        \
        case 5: SNAG_MNU_M(KERNEL, 5, M, grid, st, __VA_ARGS__); break;        \
        default: SNAG_MNU_M(KERNEL, 8, M, grid, st, __VA_ARGS__); break;       \
    }
#define SNAG_NU(KERNEL, D, grid, st, ...)                                                          \
    switch (pick_nu(D)) {                                                                          \
        case 2: hipLaunchKernelGGL((KERNEL<2>), grid, dim3(256), 0, st, __VA_ARGS__); break;       \
        case 5: hipLaunchKernelGGL((KERNEL<5>), grid, dim3(256), 0, st, __VA_ARGS__); break;       \
        default: hipLaunchKernelGGL((KERNEL<8>), grid, dim3(256), 0, st, __VA_ARGS__); break;      \
    }

static snag_gemm_desc gemm_desc(int M, int N, int K, const float* A, int64_t lda, bool akc, const float* B, int64_t ldb,
                                bool bkc, float* C, int64_t ldc) {
    snag_gemm_desc d{};
    d.M = M; d.N = N; d.K = K;
    d.A = A; d.lda = lda; d.a_kcontig = akc;
    d.B = B; d.ldb = ldb; d.b_kcontig = bkc;
    d.C = C; d.ldc = ldc;
    d.alpha = 1.0f;
    return d;
}

#ifndef SNAG_FULL_BWD_W_ROUND
#define SNAG_FULL_BWD_W_ROUND 512        // workgroups the full layer's two weight gradients (one grouped launch) are split into
#endif
#ifndef SNAG_RO_BWD_W_ROUND
#define SNAG_RO_BWD_W_ROUND 256          // workgroups the weight gradients of the merged backward launch are split into (0: the K split a launch of their own would pick; A/B)
#endif
#ifndef SNAG_RO_BWD_ONE_GROUP
#define SNAG_RO_BWD_ONE_GROUP 1      // the read-out layer's backward: weight gradients and input gradients as ONE grouped launch (0: two; A/B)
#endif
static constexpr int kGroupMaxProblems = SNAG_GEMM_GROUP_MAX;       // problems one grouped launch takes
static constexpr int kRedBlocks = 1024;  // fixed partial-sum fan-in for LayerNorm / bias / weight_raw gradients

static int check_fusion(const snag_fusion_desc* f) {
    SNAG_REQUIRE(f != nullptr, SNAG_ERR_ARG, "fusion: null descriptor");
    SNAG_REQUIRE(f->N >= 0 && f->M >= 1 && f->D >= 1 && f->H >= 1 && f->L >= 1, SNAG_ERR_ARG,
                 "fusion: bad sizes N=%d M=%d D=%d H=%d L=%d", f->N, f->M, f->D, f->H, f->L);
    SNAG_REQUIRE(f->M <= 6, SNAG_ERR_UNSUPPORTED, "fusion: M=%d modalities > 6", f->M);
    SNAG_REQUIRE(f->D <= 512, SNAG_ERR_UNSUPPORTED, "fusion: hidden_size=%d > 512", f->D);
    // the attention kernels keep an entity's H*M*M entries in three registers per lane (192: covers the reference's default
    // num_attention_heads = 5, config.py:108, with all six modalities = 180)
    SNAG_REQUIRE(f->H * f->M * f->M <= kMaxAttnEntries, SNAG_ERR_UNSUPPORTED, "fusion: H*M*M=%d attention entries per entity > %d", f->H * f->M * f->M, kMaxAttnEntries);
    // same condition as the reference's assert at model/SNAG_tools.py:161
    SNAG_REQUIRE(f->D % f->H == 0, SNAG_ERR_ARG, "fusion: hidden_size %d not divisible by num_attention_heads %d", f->D, f->H);
    SNAG_REQUIRE(!f->use_intermediate || f->I >= 1, SNAG_ERR_ARG, "fusion: use_intermediate needs intermediate_size >= 1");
    SNAG_REQUIRE(f->drop_p >= 0.f && f->drop_p < 1.f, SNAG_ERR_ARG, "fusion: drop_p=%g outside [0,1)", f->drop_p);
    SNAG_REQUIRE(f->readout_slot_p1 >= 0 && f->readout_slot_p1 <= f->M, SNAG_ERR_ARG, "fusion: readout_slot_p1=%d outside [0, M=%d]", f->readout_slot_p1, f->M);
    SNAG_REQUIRE(f->readout_slot_p1 == 0 || !(f->joint || f->joint_fz || f->weight_norm), SNAG_ERR_UNSUPPORTED,
                 "fusion: the read-out-only last layer computes attention for one token only; joint / joint_fz / weight_norm need all of it");
    SNAG_REQUIRE(!f->fold_others || (f->readout_slot_p1 > 0 && f->L == 1), SNAG_ERR_UNSUPPORTED, "fusion: fold_others needs the read-out mode with one layer");
    SNAG_REQUIRE(f->n_co_fwd >= 0 && f->n_co_fwd <= 2 && (f->n_co_fwd == 0 || f->fold_others), SNAG_ERR_ARG, "fusion: n_co_fwd=%d (0..2, fold_others only)", f->n_co_fwd);
    for (int i = 0; i < f->n_co_fwd; ++i) {
        SNAG_REQUIRE(f->co_fwd[i] != nullptr, SNAG_ERR_ARG, "fusion: null co_fwd[%d]", i);
        const int rc = gemm_check(f->co_fwd[i]);
        if (rc) return rc;
    }
    if (f->N == 0) return SNAG_OK;
    SNAG_REQUIRE(f->layers && f->weight_raw && f->x0 && f->save, SNAG_ERR_ARG, "fusion: null layers / weight_raw / x0 / save");
    const int64_t need = snag_fusion_save_floats(f->N, f->M, f->D, f->H, f->I, f->L, f->use_intermediate);
    SNAG_REQUIRE(f->save_floats >= need, SNAG_ERR_ARG, "fusion: save buffer holds %ld floats, need %ld", (long)f->save_floats, (long)need);
    for (int l = 0; l < f->L; ++l) {
        const snag_fusion_layer& P = f->layers[l];
        SNAG_REQUIRE(P.wqkv && P.bqkv && P.wo && P.bo && P.ln1_g && P.ln1_b, SNAG_ERR_ARG, "fusion: layer %d has a null attention parameter", l);
        SNAG_REQUIRE(!f->use_intermediate || (P.w1 && P.b1 && P.w2 && P.b2 && P.ln2_g && P.ln2_b), SNAG_ERR_ARG,
                     "fusion: layer %d has a null FFN parameter", l);
    }
    return SNAG_OK;
}

// The read-out layer's forward as ONE launch (readout_fused.hip) when the shape is the one that kernel is written for: hidden_size 128,
// three tokens of which two are projected tokens whose keys / values the caller computes from folded weights (co_fwd), no FFN.
// Returns 1 -- nothing launched -- otherwise; the three-launch path below gives the same bits.
static int readout_one_launch_args(const snag_fusion_desc* f, const snag_fusion_layer& P, const LayerSave& S, const float* xin, RoFusedArgs& a) {
    const int N = f->N, M = f->M, D = f->D, H = f->H, g = f->readout_slot_p1 - 1;
    if (!(f->fold_others && f->n_co_fwd == 2 && M == 3 && D == 128 && !f->use_intermediate && H <= kRoMaxHeads && ((N >= SNAG_RO_FWD_MIN_ROWS && ro_rounds_filled(N)) || f->force_one_launch))) return 1;
    const int64_t MD = (int64_t)M * D;
    auto aligned = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!aligned(xin) || !aligned(P.wqkv) || !aligned(P.wo)) return 1;
    a = RoFusedArgs{};
    a.N = N; a.H = H; a.g = g; a.ldw = D;
    a.inv_sqrt_d = 1.0f / sqrtf((float)(D / H));
    a.panel[0] = xin + (int64_t)g * D; a.ldp[0] = MD;
    float* const qkv_g = S.qkv + (int64_t)g * 3 * D;
    // the schedule: q, the read-out token's key, the other keys (the one whose panel the first value needs last), the values in token order
    // (attn_ro_fwd_kernel sums p v over the tokens in that order), the output projection
    RoTile qkv_t[3][3];          // [token][q / k / v]
    int panel_of[3];
    for (int i = 0; i < 3; ++i) qkv_t[g][i] = RoTile{P.wqkv + (int64_t)i * D * D, P.bqkv + i * D, qkv_g + i * D, 3 * MD, g, i, 0};
    panel_of[g] = 0;
    int seen = 1 << g;
    for (int i = 0; i < 2; ++i) {
        const snag_gemm_desc* c = f->co_fwd[i];
        const int64_t off = c->C - (S.qkv + D);
        const bool plain = c->M == N && c->N == 2 * D && c->K == D && c->a_kcontig && c->b_kcontig && c->ldb == D && c->alpha == 1.0f && c->act == 0 &&
                           !c->pre_act && !c->keep_mask && !c->add1 && !c->add2 && !c->accumulate && !c->mirror && !c->c_transposed && c->ldc == 3 * MD &&
                           c->lda % 4 == 0 && aligned(c->A) && aligned(c->B);
        if (!plain || off < 0 || off % (3 * D) != 0 || off / (3 * D) >= M || (seen >> (off / (3 * D))) & 1) return 1;
        const int tok = (int)(off / (3 * D));
        seen |= 1 << tok;
        a.panel[1 + i] = c->A; a.ldp[1 + i] = c->lda;
        panel_of[tok] = 1 + i;
        qkv_t[tok][1] = RoTile{c->B, c->bias, c->C, c->ldc, tok, 1, 1 + i};
        qkv_t[tok][2] = RoTile{c->B + (int64_t)D * D, c->bias ? c->bias + D : nullptr, c->C + D, c->ldc, tok, 2, 1 + i};
    }
    int others[2], no = 0;
    for (int j = 0; j < 3; ++j)
        if (j != g) others[no++] = j;
    const int first_v = 0;                                               // values run over tokens 0, 1, 2
    if (others[0] == first_v) std::swap(others[0], others[1]);           // its key product last: the panel stays
    a.t[0] = qkv_t[g][0];
    a.t[1] = qkv_t[g][1];
    a.t[2] = qkv_t[others[0]][1];
    a.t[3] = qkv_t[others[1]][1];
    for (int j = 0; j < 3; ++j) a.t[4 + j] = qkv_t[j][2];
    a.t[7] = RoTile{P.wo, P.bo, S.t1, D, g, 3, 3};
    (void)panel_of;
    a.keep_probs = P.keep_probs; a.kscale = 1.0f / (1.0f - f->drop_p);
    a.probs = S.probs; a.ctx = S.ctx;
    a.keep_out = P.keep_attn_out ? P.keep_attn_out + (int64_t)g * D : nullptr; a.ldkeep = MD;
    a.resid = xin + (int64_t)g * D; a.ldres = MD;
    a.ln_g = P.ln1_g; a.ln_b = P.ln1_b; a.eps = 1e-12f;
    a.hidden = f->hidden; a.stats = S.st1;
    a.wimg = reinterpret_cast<unsigned char*>(f->save + SaveLayout(N, M, D, H, f->I, false).per_layer * f->L);
    return SNAG_OK;
}

static int fusion_fwd_readout_one_launch(const snag_fusion_desc* f, const snag_fusion_layer& P, const LayerSave& S, const float* xin, hipStream_t st) {
    RoFusedArgs a;
    const int rc = readout_one_launch_args(f, P, S, xin, a);
    return rc ? rc : readout_fwd_fused(a, st, f->weight_images_ready != 0);
}

// Last layer when only token g = readout_slot_p1 - 1 is consumed (joint_way Mformer_hd_graph): keys / values for all M tokens, but
// the query, the attention row, the output projection, the LayerNorm (and the FFN) for token g only.  hidden is then [N, D].
// Compact [N, D] activations reuse the first N*D floats of the [N*M, D] save regions.
static int fusion_fwd_readout(const snag_fusion_desc* f, const snag_fusion_layer& P, const LayerSave& S, const float* xin, hipStream_t st) {
    const int N = f->N, M = f->M, D = f->D, H = f->H, I = f->I, g = f->readout_slot_p1 - 1;
    const bool ffn = f->use_intermediate != 0;
    const int64_t NM = (int64_t)N * M, MD = (int64_t)M * D;
    const float kscale = 1.0f / (1.0f - f->drop_p);
    const dim3 egrid((unsigned)ceil_div(N, 4));
    int rc;
    if ((rc = fusion_fwd_readout_one_launch(f, P, S, xin, st)) != 1) return rc;
    // K,V of every token: columns D..3D of the stacked projection (fold_others: of the read-out token only -- the caller has written
    // the other tokens' keys / values, computed from what their tokens are projected from)
    snag_gemm_desc gd = f->fold_others ? gemm_desc(N, 2 * D, D, xin + (int64_t)g * D, MD, true, P.wqkv + (int64_t)D * D, D, true, S.qkv + (int64_t)g * 3 * D + D, 3 * MD)
                                       : gemm_desc((int)NM, 2 * D, D, xin, D, true, P.wqkv + (int64_t)D * D, D, true, S.qkv + D, 3 * D);
    gd.bias = P.bqkv + D;
    // Q of token g only: rows n*M + g of x (stride M*D) -> the q part of the same rows of qkv.  Disjoint outputs: one grouped launch.
    snag_gemm_desc gq = gemm_desc(N, D, D, xin + (int64_t)g * D, MD, true, P.wqkv, D, true, S.qkv + (int64_t)g * 3 * D, 3 * MD);
    gq.bias = P.bqkv;
    if (f->fold_others) {
        // the read-out token's q | k | v in ONE product (N x 3D: the rows' 3D columns are contiguous in the activation), grouped with the
        // caller's key / value products of the other tokens
        snag_gemm_desc gqkv = gemm_desc(N, 3 * D, D, xin + (int64_t)g * D, MD, true, P.wqkv, D, true, S.qkv + (int64_t)g * 3 * D, 3 * MD);
        gqkv.bias = P.bqkv;
        const snag_gemm_desc* grp[3] = {f->n_co_fwd > 0 ? f->co_fwd[0] : nullptr, f->n_co_fwd > 1 ? f->co_fwd[1] : nullptr, nullptr};
        grp[f->n_co_fwd] = &gqkv;                                  // (the caller's products first: they are the larger ones)
        if ((rc = f->n_co_fwd > 0 ? gemm_dispatch_group(grp, f->n_co_fwd + 1, st) : gemm_dispatch(&gqkv, st))) return rc;
    } else if ((rc = gemm_dispatch_pair(&gd, &gq, st))) return rc;
    SNAG_MNU(attn_ro_fwd_kernel, M, D, egrid, st, (const float*)S.qkv, N, D, H, g, 1.0f / sqrtf((float)(D / H)), P.keep_probs, kscale, S.probs, S.ctx);
    if ((rc = check_launch("fusion attention (read-out)"))) return rc;
    gd = gemm_desc(N, D, D, S.ctx, D, true, P.wo, D, true, S.t1, D);
    gd.bias = P.bo; gd.keep_scale = kscale; gd.add1 = xin + (int64_t)g * D; gd.ld1 = MD;
    if (P.keep_attn_out) { gd.keep_mask = P.keep_attn_out + (int64_t)g * D; gd.ldmask = MD; }
    float* y1 = ffn ? S.y1 : f->hidden;
    // dense + dropout + residual AND the LayerNorm of the row in one launch where a tile holds whole rows (D <= 128)
    if ((rc = gemm_dispatch_ln(&gd, P.ln1_g, P.ln1_b, 1e-12f, y1, S.st1, st)) == 1) {
        if ((rc = gemm_dispatch(&gd, st))) return rc;
        SNAG_NU(ln_fwd_kernel, D, egrid, st, (const float*)S.t1, (int64_t)N, D, P.ln1_g, P.ln1_b, 1e-12f, y1, S.st1);
        rc = check_launch("fusion LayerNorm (read-out)");
    }
    if (rc) return rc;
    if (ffn) {
        gd = gemm_desc(N, I, D, y1, D, true, P.w1, D, true, S.g, I);
        gd.bias = P.b1; gd.act = 2; gd.pre_act = S.u;
        if ((rc = gemm_dispatch(&gd, st))) return rc;
        gd = gemm_desc(N, D, I, S.g, I, true, P.w2, I, true, S.t2, D);
        gd.bias = P.b2; gd.keep_scale = kscale; gd.add1 = y1; gd.ld1 = D;
        if (P.keep_ffn_out) { gd.keep_mask = P.keep_ffn_out + (int64_t)g * D; gd.ldmask = MD; }
        if ((rc = gemm_dispatch_ln(&gd, P.ln2_g, P.ln2_b, 1e-12f, f->hidden, S.st2, st)) == 1) {
            if ((rc = gemm_dispatch(&gd, st))) return rc;
            SNAG_NU(ln_fwd_kernel, D, egrid, st, (const float*)S.t2, (int64_t)N, D, P.ln2_g, P.ln2_b, 1e-12f, f->hidden, S.st2);
            rc = check_launch("fusion LayerNorm 2 (read-out)");
        }
        if (rc) return rc;
    }
    return SNAG_OK;
}

int fusion_fwd(const snag_fusion_desc* f, hipStream_t st) {
    const int N = f->N, M = f->M, D = f->D, H = f->H, I = f->I;
    const bool ffn = f->use_intermediate != 0;
    const int64_t NM = (int64_t)N * M;
    SNAG_REQUIRE(NM < (1LL << 31), SNAG_ERR_UNSUPPORTED, "fusion: N*M=%ld rows >= 2^31", (long)NM);
    SNAG_REQUIRE(f->hidden != nullptr, SNAG_ERR_ARG, "fusion_fwd: null hidden output");
    const SaveLayout lay(N, M, D, H, I, ffn);
    const float kscale = 1.0f / (1.0f - f->drop_p);
    const dim3 egrid((unsigned)ceil_div(N, 4)), rgrid((unsigned)ceil_div(NM, 4));
    const float* xin = f->x0;
    int rc;
    for (int l = 0; l < f->L; ++l) {
        const snag_fusion_layer& P = f->layers[l];
        const LayerSave S = lay.at(f->save, l);
        const bool last = (l == f->L - 1);
        if (last && f->readout_slot_p1 > 0) return fusion_fwd_readout(f, P, S, xin, st);
        // Q,K,V = X W^T + b                                   (:183-186)
        snag_gemm_desc g = gemm_desc((int)NM, 3 * D, D, xin, D, true, P.wqkv, D, true, S.qkv, 3 * D);
        g.bias = P.bqkv;
        if ((rc = gemm_dispatch(&g, st))) return rc;
        // softmax(QK^T/sqrt(d)), dropout, P~V                   (:192-206)
        SNAG_MNU(attn_fwd_kernel, M, D, egrid, st, (const float*)S.qkv, N, D, H, 1.0f / sqrtf((float)(D / H)), P.keep_probs, kscale, S.probs, S.ctx);
        if ((rc = check_launch("fusion attention"))) return rc;
        // dense + dropout + residual                            (:219-221)
        g = gemm_desc((int)NM, D, D, S.ctx, D, true, P.wo, D, true, S.t1, D);
        g.bias = P.bo; g.keep_mask = P.keep_attn_out; g.keep_scale = kscale; g.add1 = xin; g.ld1 = D;
        float* y1 = (!ffn && last) ? f->hidden : S.y1;
        if ((rc = gemm_dispatch_ln(&g, P.ln1_g, P.ln1_b, 1e-12f, y1, S.st1, st)) == 1) {
            if ((rc = gemm_dispatch(&g, st))) return rc;
            SNAG_NU(ln_fwd_kernel, D, rgrid, st, (const float*)S.t1, NM, D, P.ln1_g, P.ln1_b, 1e-12f, y1, S.st1);
            rc = check_launch("fusion LayerNorm");
        }
        if (rc) return rc;
        float* yout = y1;
        if (ffn) {                                              // :295-298
            g = gemm_desc((int)NM, I, D, y1, D, true, P.w1, D, true, S.g, I);
            g.bias = P.b1; g.act = 2; g.pre_act = S.u;
            if ((rc = gemm_dispatch(&g, st))) return rc;
            g = gemm_desc((int)NM, D, I, S.g, I, true, P.w2, I, true, S.t2, D);
            g.bias = P.b2; g.keep_mask = P.keep_ffn_out; g.keep_scale = kscale; g.add1 = y1; g.ld1 = D;
            yout = last ? f->hidden : S.y2;
            if ((rc = gemm_dispatch_ln(&g, P.ln2_g, P.ln2_b, 1e-12f, yout, S.st2, st)) == 1) {
                if ((rc = gemm_dispatch(&g, st))) return rc;
                SNAG_NU(ln_fwd_kernel, D, rgrid, st, (const float*)S.t2, NM, D, P.ln2_g, P.ln2_b, 1e-12f, yout, S.st2);
                rc = check_launch("fusion LayerNorm 2");
            }
            if (rc) return rc;
        }
        xin = yout;
    }
    if (f->joint || f->joint_fz || f->weight_norm) {
        const LayerSave S = lay.at(f->save, f->L - 1);
        SNAG_MNU(fuse_out_fwd_kernel, M, D, egrid, st, (const float*)S.probs, f->x0, f->weight_raw, N, D, H, f->joint, f->joint_fz, f->weight_norm);
        if ((rc = check_launch("fusion epilogue"))) return rc;
    }
    return SNAG_OK;
}

// ---- backward work-buffer carving --------------------------------------------------------------
struct Work {
    float *dA, *dB, *dC, *dqkv, *dU, *dpx, *dx0e, *part, *gemm_ws;
    int64_t gemm_ws_floats;
};

static int64_t bwd_gemm_ws(int N, int M, int D, int I, bool ffn) {
    const int NM = N * M;
    // (a sum, not a max: in read-out mode the slabs of dWo, dWq and dWkv wait side by side for ONE reduce launch)
    int64_t w = gemm_workspace_floats(3 * D, D, NM) + 64 + 2 * (gemm_workspace_floats(D, D, NM) + 64);
    if (ffn) w = std::max({w, gemm_workspace_floats(D, I, NM), gemm_workspace_floats(I, D, NM)});
    return w;
}

static int64_t work_floats(int N, int M, int D, int H, int I, bool ffn) {
    const int64_t NM = (int64_t)N * M;
    int64_t w = 3 * NM * D + NM * 3 * D + (ffn ? NM * I : 0) + (int64_t)N * H * M * M + NM * D;
    w += (int64_t)kRedBlocks * 3 * std::max(3 * D, I) + 3 * D + 64;
    w += bwd_gemm_ws(N, M, D, I, ffn);
    return (w + 3) / 4 * 4 + 64;
}

static Work carve(float* base, int N, int M, int D, int H, int I, bool ffn) {
    const int64_t NM = (int64_t)N * M;
    Work w{};
    float* p = base;
    w.dA = p; p += NM * D;
    w.dB = p; p += NM * D;
    w.dC = p; p += NM * D;
    w.dqkv = p; p += NM * 3 * D;
    if (ffn) { w.dU = p; p += NM * I; }
    w.dx0e = p; p += NM * D;
    w.dpx = p; p += (int64_t)N * H * M * M;
    p += (4 - ((p - base) & 3)) & 3;
    w.part = p; p += (int64_t)kRedBlocks * 3 * std::max(3 * D, I) + 3 * D + 64;
    p += (4 - ((p - base) & 3)) & 3;
    w.gemm_ws = p;
    w.gemm_ws_floats = bwd_gemm_ws(N, M, D, I, ffn);
    return w;
}

// bias gradient: out[c] (=|+=) sum_r x[r, c]
static int colsum(const float* x, int64_t R, int C, float* part, float* out, int accumulate, hipStream_t st) {
    const int by = (int)std::min<int64_t>(256, (R + 3) / 4);
    hipLaunchKernelGGL(colsum_part_kernel, dim3(ceil_div(C, 64), by), dim3(256), 0, st, x, R, C, (int64_t)C, part);
    hipLaunchKernelGGL(slab_colsum_kernel, dim3(ceil_div(C, 64)), dim3(1024), 0, st, (const float*)part, by, (int64_t)C, C, out, accumulate);
    return check_launch("fusion bias gradient");
}

// LayerNorm backward + the bias gradient of the dense layer in front of it (colsum of what flows into that dense)
static int ln_bwd(const float* dy, const float* t, const float* stats, const float* g, int64_t R, int D, const uint8_t* keep,
                  float kscale, float* dt, float* dtm, float* part, float* dgamma, float* dbeta, float* dbias, int accumulate,
                  hipStream_t st, int64_t keep_ld = 0, bool zero_dy = false, bool defer_reduce = false) {
    const int blocks = (int)std::min<int64_t>(kRedBlocks, (R + 3) / 4);
    SNAG_NU(ln_bwd_kernel, D, dim3(blocks), st, dy, t, stats, g, R, D, keep, keep_ld > 0 ? keep_ld : (int64_t)D, kscale, dt, dtm, part,
            zero_dy ? const_cast<float*>(dy) : (float*)nullptr);
    if (defer_reduce) return check_launch("fusion LayerNorm backward");       // the caller reduces this slab together with another one
    // part is [blocks][3][D]: one launch reduces the three rows as 3*D "columns" of stride 3*D, each into its own destination
    hipLaunchKernelGGL(slab_colsum3_kernel, dim3(ceil_div(3 * D, 64)), dim3(1024), 0, st, (const float*)part, blocks, D, dgamma, dbeta, dbias, accumulate);
    return check_launch("fusion LayerNorm backward");
}

int fusion_bwd(const snag_fusion_bwd_desc* b, hipStream_t st) {
    const snag_fusion_desc* f = &b->f;
    bool next_group_done = false;            // b->next_group went out with this call's last reduce riding in it
    // the full layer's weight-gradient reduce of layer 0, held back to ride in b->next_group at the end of the call (nothing of this call reads its results)
    snag_gemm_desc pend_g0{}, pend_g1{};
    SlabColsum pend_slabs{};
    bool pending = false;
    const int N = f->N, M = f->M, D = f->D, H = f->H, I = f->I, L = f->L;
    const bool ffn = f->use_intermediate != 0;
    const int64_t NM = (int64_t)N * M;
    const int acc = b->accumulate;
    const SaveLayout lay(N, M, D, H, I, ffn);
    const Work W = carve(f->work, N, M, D, H, I, ffn);
    const float kscale = 1.0f / (1.0f - f->drop_p);
    const dim3 egrid((unsigned)ceil_div(N, 4));
    int rc;

    // 1. epilogue backward: joint / joint_fz / weight_norm -> d_x0 (normalise routes), dP~ (weights route), d weight_raw
    const bool has_epi = b->d_joint || b->d_joint_fz || b->d_weight_norm;
    const float* dpx = nullptr;
    const float* dx0e = nullptr;
    if (has_epi) {
        const LayerSave S = lay.at(f->save, L - 1);
        const int blocks = (int)std::min<int64_t>(kRedBlocks, (N + 3) / 4);
        SNAG_MNU(fuse_out_bwd_kernel, M, D, dim3(blocks), st, (const float*)S.probs, f->x0, f->weight_raw, N, D, H, b->d_joint,
                 b->d_joint_fz, b->d_weight_norm, W.dx0e, W.dpx, W.part);
        if ((rc = check_launch("fusion epilogue backward"))) return rc;
        if (b->d_weight_raw) {
            hipLaunchKernelGGL(weight_raw_bwd_kernel, dim3(1), dim3(64), 0, st, (const float*)W.part, blocks, f->weight_raw, b->d_weight_raw, acc);
            if ((rc = check_launch("fusion weight_raw backward"))) return rc;
        }
        dpx = W.dpx;
        dx0e = W.dx0e;
    } else if (b->d_weight_raw && !acc) {
        if (hipMemsetAsync(b->d_weight_raw, 0, 6 * sizeof(float), st) != hipSuccess) return fail(SNAG_ERR_LAUNCH, "fusion_bwd: memset failed");
    }

    // 2. layers, last to first.  dout = gradient w.r.t. the layer's output ([N, D] in read-out mode).
    const float* dout = b->d_hidden;
    if (dout == nullptr) {
        if (hipMemsetAsync(W.dC, 0, NM * D * sizeof(float), st) != hipSuccess) return fail(SNAG_ERR_LAUNCH, "fusion_bwd: memset failed");
        dout = W.dC;
    }
    // the full layers read d_hidden through ln_bwd as well: its first call clears it when asked to
    bool zero_up_full = b->zero_d_hidden != 0 && b->d_hidden != nullptr && f->readout_slot_p1 == 0;
    for (int l = L - 1; l >= 0; --l) {
        const snag_fusion_layer& P = f->layers[l];
        const snag_fusion_layer_grad& G = b->grads[l];
        const LayerSave S = lay.at(f->save, l);
        const float* xin = (l == 0) ? f->x0 : (ffn ? lay.at(f->save, l - 1).y2 : lay.at(f->save, l - 1).y1);
        snag_gemm_desc g;
        if (l == L - 1 && f->readout_slot_p1 > 0) {
            // ---- last layer, token ro only (see fusion_fwd_readout); dout is [N, D] --------------------------------------------
            const int ro = f->readout_slot_p1 - 1;
            const int64_t MD = (int64_t)M * D;
            bool zero_up = b->zero_d_hidden != 0 && b->d_hidden != nullptr;      // the first consumer of d_hidden clears it
            if (ffn) {
                float* dtm = P.keep_ffn_out ? W.dB : nullptr;
                if ((rc = ln_bwd(dout, S.t2, S.st2, P.ln2_g, N, D, P.keep_ffn_out ? P.keep_ffn_out + (int64_t)ro * D : nullptr, kscale, W.dA, dtm, W.part,
                                 G.ln2_g, G.ln2_b, G.b2, acc, st, MD, zero_up))) return rc;
                zero_up = false;
                const float* dz = dtm ? dtm : W.dA;
                g = gemm_desc(D, I, N, dz, D, false, S.g, I, false, G.w2, I);
                g.accumulate = acc; g.workspace = W.gemm_ws; g.workspace_floats = W.gemm_ws_floats;
                if ((rc = gemm_dispatch(&g, st))) return rc;
                g = gemm_desc(N, I, D, dz, D, true, P.w2, I, false, W.dU, I);
                g.act = 3; g.aux = S.u; g.ldaux = I;
                if ((rc = gemm_dispatch(&g, st))) return rc;
                g = gemm_desc(I, D, N, W.dU, I, false, S.y1, D, false, G.w1, D);
                g.accumulate = acc; g.workspace = W.gemm_ws; g.workspace_floats = W.gemm_ws_floats;
                if ((rc = gemm_dispatch(&g, st))) return rc;
                if ((rc = colsum(W.dU, N, I, W.part, G.b1, acc, st))) return rc;
                g = gemm_desc(N, D, I, W.dU, I, true, P.w1, D, false, W.dC, D);
                g.add1 = W.dA; g.ld1 = D;
                if ((rc = gemm_dispatch(&g, st))) return rc;
                dout = W.dC;
            }
            float* dtm = P.keep_attn_out ? W.dB : nullptr;
            // (its slab [blocks][3][D] stays in W.part; the attention backward's goes behind it and ONE launch reduces both)
            // ln_bwd_done: the caller's scorer has carried the gradient through this LayerNorm already (snag_score_margin_fwd_bwd_seg_ln wrote
            // dA, dB and a slab of ceil(N / 16) rows at the offsets of snag_fusion_readout_ln_offsets)
            const bool ln_done = b->ln_bwd_done != 0 && !ffn;
            if (!ln_done && (rc = ln_bwd(dout, S.t1, S.st1, P.ln1_g, N, D, P.keep_attn_out ? P.keep_attn_out + (int64_t)ro * D : nullptr, kscale, W.dA, dtm, W.part,
                                         G.ln1_g, G.ln1_b, G.bo, acc, st, MD, zero_up, true))) return rc;
            const int lnblocks = ln_done ? ceil_div(N, 16) : (int)std::min<int64_t>(kRedBlocks, (N + 3) / 4);
            float* partq = W.part + (int64_t)kRedBlocks * 3 * D;
            const float* dz = dtm ? dtm : W.dA;
            // (dWo = dz^T ctx_g is launched further down, paired with dWq)
            // attention backward: dq_g [N, D] (behind dkv in the dqkv region), dkv [N*M, 2D]
            float* dkv = W.dqkv;
            float* dq_g = W.dqkv + NM * 2 * D;
            int ablocks = (int)std::min<int64_t>(kRedBlocks, (N + 3) / 4);
            // fold_others with M >= 3: the read-out token's dq | dk | dv side by side in the [N, 3D] block where dq_g alone would go
            float* ro3 = (f->fold_others && M >= 3) ? dq_g : nullptr;
            const bool one_launch = SNAG_RO_BWD_FUSED && ((N >= SNAG_RO_BWD_MIN_ROWS && ro_rounds_filled(N)) || f->force_one_launch) && !b->separate_attention_backward && ro3 && M == 3 && D == 128 && !ffn && ceil_div(N, 32) <= kRedBlocks && ((uintptr_t)dz & 15) == 0;
            if (one_launch) {
                // d ctx = dz Wo and the attention backward in ONE launch (readout_fused.hip): d ctx never travels; the bias slab has one row per
                // 32 entities (its column sums are associated differently from the separate launches': dbqkv agrees to rounding, all else bit for bit)
                RoBwdArgs a{};
                a.dz = dz; a.lddz = D; a.wo = P.wo; a.ldw = D;
                a.wimg = reinterpret_cast<unsigned char*>(f->save + SaveLayout(N, M, D, H, f->I, false).per_layer * f->L) + (size_t)64 * 6 * 128 * 16;
                a.images_ready = b->wo_images_ready != 0 && L == 1;
                a.qkv = S.qkv; a.keep_probs = P.keep_probs; a.kscale = kscale; a.inv_sqrt_d = 1.0f / sqrtf((float)(D / H));
                a.N = N; a.H = H; a.g = ro; a.dq_g = dq_g; a.dkv = dkv; a.ro3 = ro3; a.bias_part = partq;
                if ((rc = readout_bwd_attn_fused(a, st))) return rc;
                ablocks = ceil_div(N, 32);
            } else {
                // dctx_g = dz Wo -> dC
                g = gemm_desc(N, D, D, dz, D, true, P.wo, D, false, W.dC, D);
                if ((rc = gemm_dispatch(&g, st))) return rc;
                SNAG_MNU(attn_ro_bwd_kernel, M, D, dim3(ablocks), st, (const float*)S.qkv, (const float*)W.dC, N, D, H, ro, 1.0f / sqrtf((float)(D / H)),
                         P.keep_probs, kscale, dq_g, dkv, partq, ro3);
            }
            // (the two bias / LayerNorm slab sums ride in the weight gradients' reduce launch below: nobody reads them before the optimiser)
            const SlabColsum slabs{W.part, lnblocks, D, G.ln1_g, G.ln1_b, G.bo, partq, ablocks, G.bqkv, acc};
            if ((rc = check_launch("fusion attention backward (read-out)"))) return rc;
            float* dxin = (l == 0) ? b->d_x0 : W.dC;
            if (ro3) {
                // weight gradients: dWo = dz^T ctx_g [D, D] and dWqkv = [dq | dk | dv]^T x_g [3D, D] (all of wqkv: the caller adds the other
                // tokens' share to the key / value rows afterwards), with the caller's co_dw problems -- one grouped launch, one reduce
                snag_gemm_desc g0 = gemm_desc(D, D, N, dz, D, false, S.ctx, D, false, G.wo, D);
                snag_gemm_desc g1 = gemm_desc(3 * D, D, N, ro3, 3 * D, false, xin + (int64_t)ro * D, MD, false, G.wqkv, D);
                const int64_t pw = (gemm_workspace_floats(D, D, (int)NM) + 64) / 64 * 64;
                g0.accumulate = acc; g0.workspace = W.gemm_ws; g0.workspace_floats = pw;
                g1.accumulate = acc; g1.workspace = W.gemm_ws + pw; g1.workspace_floats = W.gemm_ws_floats - pw;
                g0.defer_reduce = g1.defer_reduce = 1;
                const snag_gemm_desc* grp[4] = {&g1, &g0, nullptr, nullptr};
                int ng = 2;
                for (int i = 0; i < b->n_co_dw; ++i) grp[ng++] = b->co_dw[i];
                // d(layer input), rows of token ro: [dq | dk | dv] Wqkv + dT1 (K = 3D), with the caller's co_dx problems
                g = gemm_desc(N, D, 3 * D, ro3, 3 * D, true, P.wqkv, D, false, dxin + (int64_t)ro * D, MD);
                g.add1 = W.dA; g.ld1 = D;
                if (b->d_slot_out != nullptr && l == 0) {
                    g.mirror = b->d_slot_out; g.ldmirror = b->d_slot_ld; g.mirror_rowmask = b->d_slot_rowmask; g.mirror_scale = b->d_slot_scale;
                }
                const snag_gemm_desc* gx[3] = {&g, nullptr, nullptr};
                int nx = 1;
                for (int i = 0; i < b->n_co_dx; ++i) gx[nx++] = b->co_dx[i];
#if SNAG_RO_BWD_ONE_GROUP
                // Weight gradients and input gradients depend on the attention backward alone: ONE grouped launch (the A operand's layout per
                // problem: gemm_tile_group_mixed_kernel), the weight gradients -- the XCD-ordered front of the launch -- then their one reduce
                const snag_gemm_desc* all[kGroupMaxProblems];
                int na = 0;
#if SNAG_RO_BWD_W_ROUND > 0
                // K splits of the weight gradients INSIDE this launch: the input gradients' ~3 N / 64 workgroups fill the chip anyway, so the weight
                // gradients want long k-ranges (one round of ~256 workgroups over all their 64 x 128 tiles) instead of the ~1,000 short ones a
                // launch of their own needs: 896 -> 256 workgroups at N = 15,000, launch 64.6 -> 56.6 us and its reduce 11.9 -> 9.8 us
                // (13 / 20 splits: 60.2 / 61.3; 10: 72 -- the long workgroups become the tail).  The caller's co_dw problems carry the caller's request.
                {
                    int wt = 0;
                    for (int i = 0; i < ng; ++i) wt += ceil_div(grp[i]->M, 64) * ceil_div(grp[i]->N, 128);
                    const int want = std::max(1, SNAG_RO_BWD_W_ROUND / std::max(1, wt));
                    if (g0.splits == 0) g0.splits = want;
                    if (g1.splits == 0) g1.splits = want;
                }
#endif
                for (int i = 0; i < ng; ++i) all[na++] = grp[i];
                for (int i = 0; i < nx; ++i) all[na++] = gx[i];
                if ((rc = gemm_dispatch_group(all, na, st))) return rc;
                if (b->n_next_group > 0 && l == 0) {
                    // the caller's next grouped launch goes out here, this layer's reduce (+ slab column sums) riding on the CUs it leaves idle
                    const GroupTail tail{grp, ng, &slabs};
                    if ((rc = gemm_group_checked(b->next_group, b->n_next_group, st, &tail))) return rc;
                    next_group_done = true;
                } else if ((rc = gemm_reduce_multi(grp, ng, st, nullptr, 0, &slabs))) return rc;
#else
                if ((rc = gemm_dispatch_group(grp, ng, st))) return rc;
                if ((rc = gemm_reduce_multi(grp, ng, st, nullptr, 0, &slabs))) return rc;
                if ((rc = nx > 1 ? gemm_dispatch_group(gx, nx, st) : gemm_dispatch(&g, st))) return rc;
#endif
                dout = W.dC;
                continue;
            }
            // dWo = dz^T ctx_g and dWq = dq_g^T x_g (rows 0..D of wqkv): two D x D outputs over N rows -> ONE launch (+ one reduce)
            {
                snag_gemm_desc g0 = gemm_desc(D, D, N, dz, D, false, S.ctx, D, false, G.wo, D);
                snag_gemm_desc g1 = gemm_desc(D, D, N, dq_g, D, false, xin + (int64_t)ro * D, MD, false, G.wqkv, D);
                // dWkv = dkv^T x (rows D..3D).  All three K splits leave their slabs in their own parts of the scratch; one launch reduces them
                snag_gemm_desc gk = f->fold_others     // (the read-out token's rows only: the caller owns the other tokens' share)
                                        ? gemm_desc(2 * D, D, N, dkv + (int64_t)ro * 2 * D, 2 * MD, false, xin + (int64_t)ro * D, MD, false, G.wqkv + (int64_t)D * D, D)
                                        : gemm_desc(2 * D, D, (int)NM, dkv, 2 * D, false, xin, D, false, G.wqkv + (int64_t)D * D, D);
                const int64_t pw = (gemm_workspace_floats(D, D, (int)NM) + 64) / 64 * 64;
                g0.accumulate = acc; g0.workspace = W.gemm_ws; g0.workspace_floats = pw;
                g1.accumulate = acc; g1.workspace = W.gemm_ws + pw; g1.workspace_floats = pw;
                gk.accumulate = acc; gk.workspace = W.gemm_ws + 2 * pw; gk.workspace_floats = W.gemm_ws_floats - 2 * pw;
                g0.defer_reduce = g1.defer_reduce = gk.defer_reduce = 1;
                if ((rc = gemm_dispatch_pair(&g0, &g1, st))) return rc;
                if ((rc = gemm_dispatch(&gk, st))) return rc;
                const snag_gemm_desc* three[3] = {&g0, &g1, &gk};
                if ((rc = gemm_reduce_multi(three, 3, st, nullptr, 0, &slabs))) return rc;
            }
            // d(layer input) = dkv Wkv for every token, + dq_g Wq + dT1 into the rows of token ro
            g = f->fold_others ? gemm_desc(N, D, 2 * D, dkv + (int64_t)ro * 2 * D, 2 * MD, true, P.wqkv + (int64_t)D * D, D, false, dxin + (int64_t)ro * D, MD)
                               : gemm_desc((int)NM, D, 2 * D, dkv, 2 * D, true, P.wqkv + (int64_t)D * D, D, false, dxin, D);
            if ((rc = gemm_dispatch(&g, st))) return rc;
            g = gemm_desc(N, D, D, dq_g, D, true, P.wqkv, D, false, dxin + (int64_t)ro * D, MD);
            g.add1 = W.dA; g.ld1 = D; g.accumulate = 1;
            if (b->d_slot_out != nullptr && l == 0) {          // this launch writes the final token-ro rows of d_x0: mirror them (entity-table gradient)
                g.mirror = b->d_slot_out; g.ldmirror = b->d_slot_ld; g.mirror_rowmask = b->d_slot_rowmask; g.mirror_scale = b->d_slot_scale;
            }
            if ((rc = gemm_dispatch(&g, st))) return rc;
            // (fold_others with fewer than three tokens: the caller's problems one by one)
            for (int i = 0; i < b->n_co_dw; ++i)
                if ((rc = gemm_dispatch(b->co_dw[i], st))) return rc;
            if (b->n_co_dw > 0 && (rc = gemm_reduce_multi(b->co_dw, b->n_co_dw, st))) return rc;
            for (int i = 0; i < b->n_co_dx; ++i)
                if ((rc = gemm_dispatch(b->co_dx[i], st))) return rc;
            dout = W.dC;
            continue;
        }
        if (ffn) {
            // LN2 backward -> dA = dT2 (residual route to y1), dB = dT2 * keep (route into output.dense)
            float* dtm = P.keep_ffn_out ? W.dB : nullptr;
            if ((rc = ln_bwd(dout, S.t2, S.st2, P.ln2_g, NM, D, P.keep_ffn_out, kscale, W.dA, dtm, W.part, G.ln2_g, G.ln2_b, G.b2, acc, st, 0, zero_up_full))) return rc;
            zero_up_full = false;
            const float* dz = dtm ? dtm : W.dA;
            // dW2 = dz^T G ; db2 = colsum(dz)
            g = gemm_desc(D, I, (int)NM, dz, D, false, S.g, I, false, G.w2, I);
            g.accumulate = acc; g.workspace = W.gemm_ws; g.workspace_floats = W.gemm_ws_floats;
            if ((rc = gemm_dispatch(&g, st))) return rc;
            // dU = (dz W2) * gelu'(u)
            g = gemm_desc((int)NM, I, D, dz, D, true, P.w2, I, false, W.dU, I);
            g.act = 3; g.aux = S.u; g.ldaux = I;
            if ((rc = gemm_dispatch(&g, st))) return rc;
            // dW1 = dU^T y1 ; db1
            g = gemm_desc(I, D, (int)NM, W.dU, I, false, S.y1, D, false, G.w1, D);
            g.accumulate = acc; g.workspace = W.gemm_ws; g.workspace_floats = W.gemm_ws_floats;
            if ((rc = gemm_dispatch(&g, st))) return rc;
            if ((rc = colsum(W.dU, NM, I, W.part, G.b1, acc, st))) return rc;
            // dy1 = dU W1 + dT2   -> dC
            g = gemm_desc((int)NM, D, I, W.dU, I, true, P.w1, D, false, W.dC, D);
            g.add1 = W.dA; g.ld1 = D;
            if ((rc = gemm_dispatch(&g, st))) return rc;
            dout = W.dC;
        }
        // LN1 backward -> dA = dT1 (residual route to the layer input), dB = dT1 * keep.  Its slab [blocks][3][D] stays in W.part; the
        // attention backward's goes behind it and ONE launch reduces both (as in the read-out layer)
        float* dtm = P.keep_attn_out ? W.dB : nullptr;
        if ((rc = ln_bwd(dout, S.t1, S.st1, P.ln1_g, NM, D, P.keep_attn_out, kscale, W.dA, dtm, W.part, G.ln1_g, G.ln1_b, G.bo, acc, st, 0, zero_up_full, true))) return rc;
        zero_up_full = false;
        const int lnblocks = (int)std::min<int64_t>(kRedBlocks, (NM + 3) / 4);
        float* partq = W.part + (int64_t)kRedBlocks * 3 * D;
        const float* dz = dtm ? dtm : W.dA;
        // dctx = dz Wo   -> dC  (dout is dead from here on)
        g = gemm_desc((int)NM, D, D, dz, D, true, P.wo, D, false, W.dC, D);
        if ((rc = gemm_dispatch(&g, st))) return rc;
        // attention backward -> dqkv
        const int ablocks = (int)std::min<int64_t>(kRedBlocks, (N + 3) / 4);
        SNAG_MNU(attn_bwd_kernel, M, D, dim3(ablocks), st, (const float*)S.qkv, (const float*)W.dC, N, D, H, 1.0f / sqrtf((float)(D / H)), P.keep_probs,
                 kscale, (l == L - 1) ? dpx : (const float*)nullptr, W.dqkv, partq);
        const SlabColsum slabs{W.part, lnblocks, D, G.ln1_g, G.ln1_b, G.bo, partq, ablocks, G.bqkv, acc};      // (ride in the reduce launch below)
        if ((rc = check_launch("fusion attention backward"))) return rc;
        // the layer's two weight gradients, dWqkv = dqkv^T xin [3D, D] and dWo = dz^T ctx [D, D] (both K = N M rows, independent): ONE
        // grouped launch + ONE reduce (they used to be two launches and two reduces around the attention backward)
        {
            snag_gemm_desc g0 = gemm_desc(D, D, (int)NM, dz, D, false, S.ctx, D, false, G.wo, D);
            snag_gemm_desc g1 = gemm_desc(3 * D, D, (int)NM, W.dqkv, 3 * D, false, xin, D, false, G.wqkv, D);
            const int64_t pw = (gemm_workspace_floats(D, D, (int)NM) + 64) / 64 * 64;
            g0.accumulate = acc; g0.workspace = W.gemm_ws; g0.workspace_floats = pw;
            g1.accumulate = acc; g1.workspace = W.gemm_ws + pw; g1.workspace_floats = W.gemm_ws_floats - pw;
            g0.defer_reduce = g1.defer_reduce = 1;
            // K splits for the GROUP: ~512 workgroups over all eight (3D + D) / 64 tiles together.  Planned one by one the two problems
            // ask for ~1,000 workgroups each (166 splits at 45,000 rows: 43 MB of slabs): launch 56.5 -> 48.4 us, its reduce 13.5 -> 9.5
            // (32 splits: 59.4 + 9.0; 96: 47.4 + 10.8)
            g0.splits = g1.splits = std::max(1, SNAG_FULL_BWD_W_ROUND / std::max(1, ceil_div(3 * D, 64) + ceil_div(D, 64)) / std::max(1, ceil_div(D, 128)));
            const snag_gemm_desc* grp[2] = {&g1, &g0};
            if ((rc = gemm_dispatch_group(grp, 2, st))) return rc;
            if (b->n_next_group > 0 && l == 0) {
                pend_g0 = g0; pend_g1 = g1; pend_slabs = slabs; pending = true;
            } else if ((rc = gemm_reduce_multi(grp, 2, st, nullptr, 0, &slabs))) return rc;
        }
        // d(layer input) = dqkv Wqkv + dT1 (+ epilogue route at layer 0)
        float* dxin = (l == 0) ? b->d_x0 : W.dC;
        g = gemm_desc((int)NM, D, 3 * D, W.dqkv, 3 * D, true, P.wqkv, D, false, dxin, D);
        g.add1 = W.dA; g.ld1 = D;
        if (l == 0 && dx0e) { g.add2 = dx0e; g.ld2 = D; }
        if ((rc = gemm_dispatch(&g, st))) return rc;
        dout = W.dC;
    }
    // (every other path: the caller's next grouped launch behind everything of this call, as if the caller had issued it)
    if (pending) {
        const snag_gemm_desc* grp[2] = {&pend_g1, &pend_g0};
        const GroupTail tail{grp, 2, &pend_slabs};
        return gemm_group_checked(b->next_group, b->n_next_group, st, &tail);
    }
    if (b->n_next_group > 0 && !next_group_done) return gemm_group_checked(b->next_group, b->n_next_group, st, nullptr);
    return SNAG_OK;
}

}  // namespace snag

using namespace snag;

extern "C" SNAG_API int snag_fusion_fwd_prepare_reduce(const snag_fusion_desc* f, const snag_gemm_desc* const* descs, int n, int32_t* prepared, void* stream) {
    SNAG_REQUIRE(prepared != nullptr && n >= 0 && n <= 4 && (n == 0 || descs != nullptr), SNAG_ERR_ARG, "fusion_fwd_prepare_reduce: bad arguments");
    *prepared = 0;
    for (int i = 0; i < n; ++i) {
        const int rc = gemm_check(descs[i]);
        if (rc) return rc;
    }
    RoFusedArgs a;
    RoWeightImages w{};
    bool fused = false;
    if (f != nullptr && f->N > 0 && f->L == 1 && f->readout_slot_p1 > 0 && check_fusion(f) == SNAG_OK) {
        // (one layer: the read-out layer's input is x0, known before the forward runs)
        const LayerSave S = SaveLayout(f->N, f->M, f->D, f->H, f->I, f->use_intermediate != 0).at(f->save, 0);
        if (readout_one_launch_args(f, f->layers[0], S, f->x0, a) == SNAG_OK) {
            w = readout_weight_images(a);
            w.Wt = f->layers[0].wo;          // + the backward's d ctx = dz Wo operand (images 64 .. 71)
            w.n_images += 8;
            fused = true;
        }
    }
    const int rc = gemm_reduce_multi(descs, n, (hipStream_t)stream, nullptr, 0, nullptr, fused ? &w : nullptr);
    if (rc == SNAG_OK && fused) *prepared = 1;
    return rc;
}

extern "C" SNAG_API int64_t snag_fusion_save_floats(int N, int M, int D, int H, int I, int L, int use_intermediate) {
    if (N < 0 || M < 1 || D < 1 || H < 1 || L < 1) return 0;
    // (+ the split weight images of the one-launch read-out forward behind the last layer's activations -- and the backward's Wo^T: 864 KB whatever the shape)
    return SaveLayout(N, M, D, H, I, use_intermediate != 0).per_layer * L + readout_fused_image_bytes() / 4;
}

extern "C" SNAG_API int64_t snag_fusion_work_floats(int N, int M, int D, int H, int I, int L, int use_intermediate) {
    if (N < 0 || M < 1 || D < 1 || H < 1 || L < 1) return 0;
    return work_floats(N, M, D, H, I, use_intermediate != 0);
}

extern "C" SNAG_API int snag_fusion_readout_offsets(const snag_fusion_desc* f, int64_t* qkv_save_offset, int64_t* dkv_work_offset) {
    SNAG_REQUIRE(f != nullptr && qkv_save_offset && dkv_work_offset, SNAG_ERR_ARG, "fusion_readout_offsets: null argument");
    SNAG_REQUIRE(f->readout_slot_p1 > 0 && f->L >= 1, SNAG_ERR_UNSUPPORTED, "fusion_readout_offsets: read-out mode only");
    const bool ffn = f->use_intermediate != 0;
    const SaveLayout lay(f->N, f->M, f->D, f->H, f->I, ffn);
    *qkv_save_offset = lay.per_layer * (f->L - 1);                       // LayerSave::qkv is the first region of a layer
    *dkv_work_offset = 3 * (int64_t)f->N * f->M * f->D;                  // Work: dA, dB, dC, then dqkv (dkv first)
    return SNAG_OK;
}

extern "C" SNAG_API int snag_fusion_readout_ln_offsets(const snag_fusion_desc* f, int64_t* t_save_offset, int64_t* stats_save_offset, int64_t* dt_work_offset,
                                                       int64_t* dtm_work_offset, int64_t* part_work_offset) {
    SNAG_REQUIRE(f != nullptr && t_save_offset && stats_save_offset && dt_work_offset && dtm_work_offset && part_work_offset, SNAG_ERR_ARG,
                 "fusion_readout_ln_offsets: null argument");
    SNAG_REQUIRE(f->readout_slot_p1 > 0 && f->L >= 1 && !f->use_intermediate, SNAG_ERR_UNSUPPORTED, "fusion_readout_ln_offsets: read-out mode without FFN only");
    SNAG_REQUIRE(ceil_div(f->N, 16) <= kRedBlocks, SNAG_ERR_UNSUPPORTED, "fusion_readout_ln_offsets: N = %d > %d rows (the slab holds %d partial rows of 16)", f->N,
                 16 * kRedBlocks, kRedBlocks);
    const SaveLayout lay(f->N, f->M, f->D, f->H, f->I, false);
    float* const zero = nullptr;
    const LayerSave S = lay.at(zero, f->L - 1);
    const Work W = carve(zero, f->N, f->M, f->D, f->H, f->I, false);
    *t_save_offset = S.t1 - zero;
    *stats_save_offset = S.st1 - zero;
    *dt_work_offset = W.dA - zero;
    *dtm_work_offset = W.dB - zero;
    *part_work_offset = W.part - zero;
    return SNAG_OK;
}

namespace snag {
__global__ __launch_bounds__(256) void small_matmul_kernel(SmallMMSet s) { small_matmul_body(s, (int)blockIdx.x); }
}  // namespace snag

extern "C" SNAG_API int snag_small_matmul(const snag_small_mm* problems, int n_problems, void* stream) {
    SNAG_REQUIRE(problems != nullptr && n_problems >= 1, SNAG_ERR_ARG, "small_matmul: 1..4 problems");
    SmallMMSet s{};
    const int blocks = fill_small_mm(problems, n_problems, s, "small_matmul");
    if (blocks <= 0) return blocks;
    hipLaunchKernelGGL(small_matmul_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, s);
    return check_launch("snag_small_matmul");
}

extern "C" SNAG_API int snag_fusion_fwd(const snag_fusion_desc* f, void* stream) {
    int rc = check_fusion(f);
    if (rc) return rc;
    if (f->N == 0) return SNAG_OK;
    return fusion_fwd(f, (hipStream_t)stream);
}

extern "C" SNAG_API int snag_fusion_bwd(const snag_fusion_bwd_desc* b, void* stream) {
    SNAG_REQUIRE(b != nullptr, SNAG_ERR_ARG, "fusion_bwd: null descriptor");
    int rc = check_fusion(&b->f);
    if (rc) return rc;
    SNAG_REQUIRE(b->n_next_group >= 0 && b->n_next_group <= SNAG_GEMM_GROUP_MAX && (b->n_next_group == 0 || b->next_group != nullptr), SNAG_ERR_ARG,
                 "fusion_bwd: n_next_group=%d (0..%d problems)", b->n_next_group, SNAG_GEMM_GROUP_MAX);
    for (int i = 0; i < b->n_next_group; ++i) {
        SNAG_REQUIRE(b->next_group[i] != nullptr, SNAG_ERR_ARG, "fusion_bwd: null descriptor %d in next_group", i);
        if ((rc = gemm_check(b->next_group[i]))) return rc;
    }
    if (b->f.N == 0) return b->n_next_group > 0 ? snag_gemm_f32_group(b->next_group, b->n_next_group, stream) : SNAG_OK;
    SNAG_REQUIRE(b->d_x0 && b->grads && b->f.work, SNAG_ERR_ARG, "fusion_bwd: null d_x0 / grads / work");
    SNAG_REQUIRE(b->f.readout_slot_p1 == 0 || !(b->d_joint || b->d_joint_fz || b->d_weight_norm), SNAG_ERR_UNSUPPORTED,
                 "fusion_bwd: read-out-only mode takes d_hidden [N, D] alone");
    SNAG_REQUIRE(b->d_slot_out == nullptr || (b->f.readout_slot_p1 != 0 && b->f.L == 1 && b->d_slot_ld >= b->f.D), SNAG_ERR_UNSUPPORTED,
                 "fusion_bwd: d_slot_out needs read-out mode with one layer (and d_slot_ld >= D)");
    SNAG_REQUIRE(b->n_co_dx >= 0 && b->n_co_dx <= 2 && b->n_co_dw >= 0 && b->n_co_dw <= 2 && ((b->n_co_dx == 0 && b->n_co_dw == 0) || b->f.fold_others), SNAG_ERR_ARG,
                 "fusion_bwd: n_co_dx=%d n_co_dw=%d (0..2 each, fold_others only)", b->n_co_dx, b->n_co_dw);
    for (int i = 0; i < b->n_co_dx + b->n_co_dw; ++i) {
        const snag_gemm_desc* d = i < b->n_co_dx ? b->co_dx[i] : b->co_dw[i - b->n_co_dx];
        SNAG_REQUIRE(d != nullptr, SNAG_ERR_ARG, "fusion_bwd: null co-launched descriptor %d", i);
        if ((rc = gemm_check(d))) return rc;
        SNAG_REQUIRE(i < b->n_co_dx || d->defer_reduce, SNAG_ERR_ARG, "fusion_bwd: co_dw[%d] must carry defer_reduce (this call finishes it)", i - b->n_co_dx);
    }
    const int64_t need = snag_fusion_work_floats(b->f.N, b->f.M, b->f.D, b->f.H, b->f.I, b->f.L, b->f.use_intermediate);
    SNAG_REQUIRE(b->f.work_floats >= need, SNAG_ERR_ARG, "fusion_bwd: work buffer holds %ld floats, need %ld", (long)b->f.work_floats, (long)need);
    // ln_bwd_done: the caller's scorer wrote dA / dB and a slab of ceil(N / 16) partial rows where the read-out LayerNorm's backward would have;
    // the slab region holds kRedBlocks rows and the epilogue backward (d_joint / d_joint_fz / d_weight_norm) writes the same region first
    SNAG_REQUIRE(!b->ln_bwd_done || b->f.use_intermediate ||
                     (ceil_div(b->f.N, 16) <= kRedBlocks && b->f.readout_slot_p1 > 0 && !b->d_joint && !b->d_joint_fz && !b->d_weight_norm),
                 SNAG_ERR_ARG, "fusion_bwd: ln_bwd_done needs a read-out-only last layer of at most %d rows and no joint / weight_norm gradients (N = %d)",
                 16 * kRedBlocks, b->f.N);
    for (int l = 0; l < b->f.L; ++l) {
        const snag_fusion_layer_grad& G = b->grads[l];
        SNAG_REQUIRE(G.wqkv && G.bqkv && G.wo && G.bo && G.ln1_g && G.ln1_b, SNAG_ERR_ARG, "fusion_bwd: layer %d has a null attention gradient", l);
        SNAG_REQUIRE(!b->f.use_intermediate || (G.w1 && G.b1 && G.w2 && G.b2 && G.ln2_g && G.ln2_b), SNAG_ERR_ARG,
                     "fusion_bwd: layer %d has a null FFN gradient", l);
    }
    return fusion_bwd(b, (hipStream_t)stream);
}

extern "C" SNAG_API int snag_pack_tokens(const float* const* src_host, int M, int64_t N, int D, int g_slot, const float* noise,
                                         const uint8_t* noise_mask, float half_mask_ratio, float* out, void* stream) {
    SNAG_REQUIRE(src_host && out && M >= 1 && M <= 6 && N >= 0 && D >= 1, SNAG_ERR_ARG, "pack_tokens: bad arguments (M=%d)", M);
    SNAG_REQUIRE((noise == nullptr) == (noise_mask == nullptr), SNAG_ERR_ARG, "pack_tokens: noise and noise_mask go together");
    if (N == 0) return SNAG_OK;
    TokenPtrs tp{};
    for (int m = 0; m < M; ++m) {
        SNAG_REQUIRE(src_host[m] != nullptr, SNAG_ERR_ARG, "pack_tokens: token %d is null", m);
        tp.p[m] = src_host[m];
    }
    const int64_t total = N * M * (int64_t)D;
    const int blocks = (int)std::min<int64_t>(4096, (total + 255) / 256);
    hipLaunchKernelGGL(pack_tokens_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, tp, M, N, D, g_slot, noise, noise_mask,
                       half_mask_ratio, out);
    return check_launch("snag_pack_tokens");
}

extern "C" SNAG_API int snag_token_mean(const float* hidden, int64_t N, int M, int D, float* out, void* stream) {
    SNAG_REQUIRE(hidden && out && N >= 0 && M >= 1 && D >= 1, SNAG_ERR_ARG, "token_mean: bad arguments");
    if (N == 0) return SNAG_OK;
    const int64_t total = N * (int64_t)D;
    const int blocks = (int)std::min<int64_t>(4096, (total + 255) / 256);
    hipLaunchKernelGGL(token_mean_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, hidden, N, M, D, out);
    return check_launch("snag_token_mean");
}
