// The read-out layer's forward as ONE launch (host-visible part): see readout_fused.hip.
#pragma once
#include "common.hip.h"

namespace snag {

// one 64-entity x 128-column product of the chain: out[m, 0..127] = panel[m, :] . W[n, :]^T + bias[n]
struct RoTile {
    const float* W;        // 128 weight rows (nn.Linear layout [out, in]: K-contiguous), pitch RoFusedArgs::ldw floats
    const float* bias;     // [128] or null
    float* out;            // row m at out + m * ldo
    int64_t ldo;
    int token;             // k / v steps: the token whose key / value this is
    int kind;              // 0 query, 1 key, 2 value, 3 output projection
    int panel;             // the A operand: 0 .. 2 = RoFusedArgs::panel[], 3 = ctx (formed by the launch itself)
};

constexpr int kRoTiles = 8;        // q ; the three keys ; the three values in token order ; the output projection
constexpr int kRoMaxHeads = 4;

struct RoFusedArgs {
    RoTile t[kRoTiles];
    const float* panel[3];         // rows of the read-out token's input, of the two projected tokens' inputs (mid_a, mid_b): [N, 128] K-contiguous
    int64_t ldp[3];
    int ldw;
    int N, H, g;                   // entities, heads, read-out token
    float inv_sqrt_d;
    const uint8_t* keep_probs;     // [N, H, 3, 3] or null
    float kscale;
    float* probs;                  // [N, H, 3, 3] (row g written)
    float* ctx;                    // [N, 128]
    // output projection: t1 = dropout(ctx Wo^T + bo) + residual ; hidden = LayerNorm(t1) ; stats = (mean, rstd)
    const uint8_t* keep_out; int64_t ldkeep;
    const float* resid; int64_t ldres;
    const float *ln_g, *ln_b;
    float eps;
    float* hidden;
    float* stats;
    unsigned char* wimg;           // readout_fused_image_bytes() of scratch: the eight products' weights as split k-tile images (rewritten by every call)
};

int64_t readout_fused_image_bytes();

// The read-out layer's attention backward with its d ctx product in ONE launch (readout_fused.hip): d ctx = dz Wo for 32 entities per half-workgroup
// (split-bf16 products of gemm_tile_kernel's bits), then attn_ro_bwd_entity per entity with d ctx taken from the LDS row tile.
struct RoBwdArgs {
    const float* dz; int64_t lddz;       // [N, 128] K-contiguous: the gradient of the output projection's result
    const float* wo; int ldw;            // output.dense weight [128 out, 128 in]
    unsigned char* wimg;                 // 8 x 12 KB: Wo^T as split k-tile images
    int images_ready;                    // 1: they hold this step's weights already (written beside the forward's images); 0: a launch of this call writes them
    const float* qkv;
    const uint8_t* keep_probs;
    float kscale, inv_sqrt_d;
    int N, H, g;
    float *dq_g, *dkv, *ro3;
    float* bias_part;                    // [ceil(N / 32)][3][128]: per-half column sums of dq, dk, dv
};
int readout_bwd_attn_fused(const RoBwdArgs& a, hipStream_t st);
RoWeightImages readout_weight_images(const RoFusedArgs& a);
// images_ready: the weight images in a.wimg are this step's already (prepared inside another launch): no preparation launch
int readout_fwd_fused(const RoFusedArgs& a, hipStream_t st, bool images_ready);

}  // namespace snag
