// K3g — EMM prediction towers for response maps other than 16x16, on the fp32 matrix cores.
//
// The reference's second yaml family (configs/dla/DLA_34_FPN_EMM_AOT.yaml:52-63: template 7x7, search region x5)
// correlates 35x35 with 7x7: the towers of EMMPredictor.forward (EMM/feature_extractor.py:62-66: conv3x3 C->C
// without bias -> GroupNorm(32) -> ReLU, twice) then see a 29x29 map.  tower_generic_kernel (predictor.hip) is a
// scalar direct convolution for ANY shape — 3.2 ms at 30 tracks, C=128 (profiles/r02_aot_*) — this kernel is the
// same contraction as a direct implicit GEMM on v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulation):
//
//   workgroup = (track, tower, 16 output channels) x ALL HO*HO positions, 4 waves: every GroupNorm group of the tile
//               is complete inside the workgroup, so GroupNorm + affine + ReLU are fused into the epilogue;
//   wave w    = N-tiles (16 positions each) w, w+4, w+8, ...  -> up to NTW accumulator tiles in registers;
//   K order   = input-channel chunk (8) > 4 channels per MFMA (2 k-steps) > tap (9);
//   A operand = the tile's filter taps, staged per chunk as wl[ic][tap][oc]: one conflict-free ds_read_b32 per
//               (k-step, tap), reused by all the wave's N-tiles;
//   B operand = one ds_read_b32 per MFMA from zero-haloed (HO+2)^2 planes; HO is a template parameter so that every
//               (k-step, tap) offset is an immediate — no vector instruction per MFMA (VALU instructions do not
//               overlap the fp32 matrix pipe on gfx950, profiles/r02_ubench_mfma_valu_overlap.jsonl);
//   staging   = the chunk (8 contiguous planes) is fetched with coalesced buffer loads at an SGPR chunk offset into
//               registers while the previous chunk feeds the matrix cores; the LDS scatter offsets of a thread's
//               elements are computed once (the element -> (plane, row, column) map is the same for every chunk).
//   heads     = fused as in tower_wino.hip: the normalised 16-channel tile goes to zero-haloed LDS planes (over the
//               stage buffers) and v_mfma_f32_4x4x1 with block-broadcast weights (nine registers per lane) adds the
//               tile's contribution to the 4 head outputs of its tower, 64 positions per instruction group.  The
//               25.8 MB activation tensor never reaches HBM; partial sums [N][tile][4][HW] are combined in tile
//               order (+bias, ReLU on reg) by heads_combine_hw_kernel.
// Accumulation order differs from the generic kernels' (chunk > channel-in-step > tap instead of channel > tap, and
// per-tile partial head sums): same 1e-4-of-logit-scale bound against the fp64 oracle (tests/test_hip_parity.py).
#include "tower_common.h"

namespace smot {

constexpr int G_IC = 8;                           // input channels per chunk

template <int HO>
struct ConvGeom {
    static constexpr int HW = HO * HO;
    static constexpr int PW = HO + 2;
    static constexpr int PLANE = PW * PW;
    static constexpr int NTILES = (HW + 15) / 16;
    static constexpr int NTW = (NTILES + 3) / 4;                      // N-tiles per wave
    static constexpr int NLD = (G_IC * HW + 255) / 256;               // staged response elements per thread and chunk
    static constexpr int WL = G_IC * 9 * 16;                          // filter taps per chunk
    static constexpr int NWL = (WL + 255) / 256;
    static constexpr int BUF = G_IC * PLANE + WL;                     // floats per stage buffer
    static constexpr int SMEM_FLOATS = 2 * BUF + 2 * 4 * 16;          // two stage buffers + per-wave channel sums
    static constexpr int NGROUPS = (HW + 63) / 64;                    // head pass: groups of 64 positions
    static_assert(16 * PLANE <= 2 * BUF, "the head planes overlay the stage buffers");
};

// GroupNorm + affine + ReLU + fused partial heads of one (track, tower, 16-channel tile), from the convolution output in
// the direct kernel's accumulator layout: acc[t][r] = channel oc0 + 4*kq + r at position 16*(wave + 4t) + xl.  Shared by
// tower_conv_mfma_kernel (accumulators straight from its main loop) and tower_gn_heads_kernel (convolution output of
// the blocked Winograd kernel, re-loaded).  sm: 16 zero-haloed planes + 128 floats of channel sums; every wave is past its
// last use of sm when it gets here.  part_tile: where this tile's four partial head planes go.
// Zero the one-cell halo of the 16 LDS planes (every interior cell is written by the tail): 4 * (HO + 1) cells per plane
// instead of (HO + 2)^2.
template <int HO>
__device__ __forceinline__ void conv_zero_halo(float* hp, int tid) {
    using G = ConvGeom<HO>;
    for (int e = tid; e < 16 * 4 * (HO + 1); e += 256) {
        const int pl = e / (4 * (HO + 1)), c = e - pl * 4 * (HO + 1);
        const int side = c / (HO + 1), k = c - side * (HO + 1);          // four runs of HO + 1 cells around the plane
        const int y = side == 0 ? 0 : (side == 1 ? HO + 1 : (side == 2 ? 1 + k : k));
        const int x = side == 0 ? k : (side == 1 ? 1 + k : (side == 2 ? 0 : HO + 1));
        hp[pl * G::PLANE + y * G::PW + x] = 0.0f;
    }
}

// HALO_DONE: the caller zeroed the halo already (conv_zero_halo, on planes no wave still uses otherwise) — the loop and
// its barrier are skipped (tower_gn_heads_kernel does it in the shadow of its loads).
template <int HO, bool HALO_DONE = false>
__device__ __forceinline__ void conv_tile_tail(f32x4 (&acc)[ConvGeom<HO>::NTW], const int (&boff)[ConvGeom<HO>::NTW],
                                               int my_tiles, const float (&hwv)[9], const TowerParams& P, int tower,
                                               int oc0, int cpg, float eps, float* sm, float* chs,
                                               float* __restrict__ dst, int tid, int lane, int wave, int kq, int xl,
                                               long long* tr = nullptr) {
    using G = ConvGeom<HO>;
    // phase trace (smot_debug_trace): s_memtime stamps 1 .. 5 of this workgroup, or nothing
#define CT_TRACE(SLOT) \
    if (tr && tid == 0) tr[SLOT] = (long long)__builtin_amdgcn_s_memtime();
    // ---- GroupNorm (two-pass, fp32) + affine + ReLU -------------------------------------------------------------
    // acc[t][r] = conv output of channel oc0 + 4*kq + r at position 16*(wave + 4t) + xl
    bool valid[G::NTW];
#pragma unroll
    for (int t = 0; t < G::NTW; ++t)
        valid[t] = (t < G::NTW - 1 || t < my_tiles) && (16 * (wave + 4 * t) + xl < G::HW);
    const float inv_cnt = 1.0f / (float)(cpg * G::HW);
    float mean[4], rstd[4];
    {
        float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < G::NTW; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[r] += valid[t] ? acc[t][r] : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s[r] = group16_sum(s[r]);
            if (xl == 0) chs[wave * 16 + 4 * kq + r] = s[r];
        }
        __syncthreads();
        CT_TRACE(1)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int g0 = ((4 * kq + r) / cpg) * cpg;
            float m = 0.0f;
            for (int ch = g0; ch < g0 + cpg; ++ch) m += ((chs[ch] + chs[16 + ch]) + chs[32 + ch]) + chs[48 + ch];
            mean[r] = m * inv_cnt;
        }
    }
    {
        float q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < G::NTW; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = acc[t][r] - mean[r];
                q[r] += valid[t] ? d * d : 0.0f;
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            q[r] = group16_sum(q[r]);
            if (xl == 0) chs[64 + wave * 16 + 4 * kq + r] = q[r];
        }
        __syncthreads();
        CT_TRACE(2)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int g0 = ((4 * kq + r) / cpg) * cpg;
            float v = 0.0f;
            for (int ch = g0; ch < g0 + cpg; ++ch)
                v += ((chs[64 + ch] + chs[80 + ch]) + chs[96 + ch]) + chs[112 + ch];
            rstd[r] = 1.0f / sqrtf(v * inv_cnt + eps);
        }
    }
    // ---- normalised tile -> zero-haloed LDS planes (over the stage buffers: every wave is past its last MFMA) -----
    float* hp = sm;
    if constexpr (!HALO_DONE) {
        conv_zero_halo<HO>(hp, tid);
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int oc = 4 * kq + r;
        const float ga = P.gamma[tower][oc0 + oc], be = P.beta[tower][oc0 + oc];
#pragma unroll
        for (int t = 0; t < G::NTW; ++t)
            if (valid[t]) {
                const float v = (acc[t][r] - mean[r]) * rstd[r] * ga + be;
                hp[oc * G::PLANE + (boff[t] - kq * G::PLANE) + G::PW + 1] = relu_nan(v);
            }
    }
    __syncthreads();
    CT_TRACE(3)

    // ---- fused partial heads: 16 channels x 9 taps -> 4 head outputs per position, 64 positions per group ---------
    // A wave's groups (g = wave, wave + 4, ...: up to NGW of them) run SIDE BY SIDE: one accumulator chain per group — a
    // single chain is 144 dependent matrix instructions (each waits for the previous one's result: 4.7 k cycles per
    // group), four chains pipeline.
    constexpr int NGW = (G::NGROUPS + 3) / 4;
    const float* pl0[NGW];
    f32x4 hacc[NGW];
#pragma unroll
    for (int q = 0; q < NGW; ++q) {
        const int pc = min(64 * (wave + 4 * q) + lane, G::HW - 1);        // (groups past the map re-read the last cell)
        const int y = pc / HO, x = pc - y * HO;
        pl0[q] = hp + y * G::PW + x;
        hacc[q] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    }
#define C_HEAD(ID)                                                                                               \
    _Pragma("unroll") for (int q = 0; q < NGW; ++q)                                                              \
        hacc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(hwv[(ID) / 16],                                             \
                                                     pl0[q][((ID) / 9) * G::PLANE + (((ID) % 9) / 3) * G::PW + ((ID) % 9) % 3], \
                                                     hacc[q], 4, (ID) % 16, 0);
#define C_HEAD16(R)                                                                                              \
    C_HEAD((R) * 16 + 0) C_HEAD((R) * 16 + 1) C_HEAD((R) * 16 + 2) C_HEAD((R) * 16 + 3) C_HEAD((R) * 16 + 4)        \
    C_HEAD((R) * 16 + 5) C_HEAD((R) * 16 + 6) C_HEAD((R) * 16 + 7) C_HEAD((R) * 16 + 8) C_HEAD((R) * 16 + 9)        \
    C_HEAD((R) * 16 + 10) C_HEAD((R) * 16 + 11) C_HEAD((R) * 16 + 12) C_HEAD((R) * 16 + 13)                       \
    C_HEAD((R) * 16 + 14) C_HEAD((R) * 16 + 15)
    C_HEAD16(0) C_HEAD16(1) C_HEAD16(2) C_HEAD16(3) C_HEAD16(4) C_HEAD16(5) C_HEAD16(6) C_HEAD16(7) C_HEAD16(8)
#undef C_HEAD16
#undef C_HEAD
    CT_TRACE(4)
#pragma unroll
    for (int q = 0; q < NGW; ++q) {
        const int p = 64 * (wave + 4 * q) + lane;
        if (wave + 4 * q < G::NGROUPS && p < G::HW) {
            dst[0 * G::HW + p] = hacc[q][0];
            dst[1 * G::HW + p] = hacc[q][1];
            dst[2 * G::HW + p] = hacc[q][2];
            dst[3 * G::HW + p] = hacc[q][3];
        }
    }
    CT_TRACE(5)
#undef CT_TRACE
}

template <int HO>
__global__ void __launch_bounds__(256, 2)
tower_conv_mfma_kernel(const float* __restrict__ resp, TowerParams P, int C, int cpg, float eps,
                       float* __restrict__ part, unsigned* __restrict__ zero_words) {
    using G = ConvGeom<HO>;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = lane >> 4, xl = lane & 15;
    const int tiles_per_tower = C >> 4;
    const int n = blockIdx.x / (2 * tiles_per_tower);
    const int rem = blockIdx.x - n * 2 * tiles_per_tower;
    const int tower = rem / tiles_per_tower;
    const int oc0 = (rem - tower * tiles_per_tower) * 16;
    const int nchunks = C / G_IC;
    if (zero_words != nullptr && rem == 0 && tid == 0) zero_words[n] = 0u;   // the decode kernel's ticket of this track

    auto make_rsrc = [](const float* base) {
        const unsigned long long a = reinterpret_cast<unsigned long long>(base);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                                 0x7fffffff, 0x00020000);
    };
    const auto rs_in = make_rsrc(resp + (size_t)n * C * G::HW);
    const auto rs_w = make_rsrc(P.w[tower] + (size_t)oc0 * C * 9);

    // head taps of this tile's channels in nine registers per lane (block b of register r = tap 16r + b, lane % 4 =
    // head output): named by the 4x4x1 instruction's A-broadcast field in the epilogue (see tower_wino.hip)
    float hwv[9];
    {
        const int o = lane & 3, blk = lane >> 2;
        const float* wsrc = (tower == 1) ? P.reg_w + (size_t)o * C * 9
                                         : (o < 2 ? P.cls_w + (size_t)o * C * 9 : P.center_w);
        const bool live = (tower == 1) || (o < 3);
        wsrc += (size_t)oc0 * 9 + blk;
#pragma unroll
        for (int r = 0; r < 9; ++r) hwv[r] = live ? wsrc[r * 16] : 0.0f;
    }

    // element -> LDS offset maps, computed once
    int ld_dst[G::NLD];                      // response element e = tid + 256 j of a chunk -> haloed plane offset (-1: none)
#pragma unroll
    for (int j = 0; j < G::NLD; ++j) {
        const int e = tid + 256 * j;
        const int ic = e / G::HW, pos = e - ic * G::HW;
        const int y = pos / HO, x = pos - y * HO;
        ld_dst[j] = (e < G_IC * G::HW) ? ic * G::PLANE + (y + 1) * G::PW + x + 1 : -1;
    }
    int wl_src[G::NWL], wl_dst[G::NWL];      // filter element -> (byte offset in the tile's filters, LDS offset)
#pragma unroll
    for (int j = 0; j < G::NWL; ++j) {
        const int e = tid + 256 * j;
        const int oc = e / (G_IC * 9), r = e - oc * (G_IC * 9);      // r = ic*9 + tap
        wl_src[j] = (e < G::WL) ? (oc * C * 9 + r) * 4 : -1;
        wl_dst[j] = G_IC * G::PLANE + r * 16 + oc;
    }
    float pre[G::NLD], prew[G::NWL];
    auto fetch = [&](int chunk) {
        const int soff = __builtin_amdgcn_readfirstlane(chunk * (G_IC * G::HW * 4));
        const int woff = __builtin_amdgcn_readfirstlane(chunk * (G_IC * 9 * 4));
#pragma unroll
        for (int j = 0; j < G::NLD; ++j)
            pre[j] = (ld_dst[j] >= 0) ? __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                                            rs_in, (unsigned)(tid + 256 * j) * 4u, soff, 0))
                                      : 0.0f;
#pragma unroll
        for (int j = 0; j < G::NWL; ++j)
            prew[j] = (wl_src[j] >= 0) ? __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_w, (unsigned)wl_src[j], woff, 0))
                                       : 0.0f;
    };
    auto stash = [&](float* buf) {
#pragma unroll
        for (int j = 0; j < G::NLD; ++j)
            if (ld_dst[j] >= 0) buf[ld_dst[j]] = pre[j];
#pragma unroll
        for (int j = 0; j < G::NWL; ++j)
            if (wl_src[j] >= 0) buf[wl_dst[j]] = prew[j];
    };

    fetch(0);
    for (int e = tid; e < 2 * G::BUF; e += 256) sm[e] = 0.0f;       // halos stay zero for the whole kernel
    __syncthreads();
    stash(sm);
    if (nchunks > 1) fetch(1);
    __syncthreads();

    // B operand base of lane (kq, xl) for the wave's t-th N-tile: channel kq of the k-step, position 16*(wave+4t)+xl
    // (positions past the end of the map re-read the last cell; their columns are never stored)
    int boff[G::NTW];
#pragma unroll
    for (int t = 0; t < G::NTW; ++t) {
        const int p = min(16 * (wave + 4 * t) + xl, G::HW - 1);
        const int y = p / HO, x = p - y * HO;
        boff[t] = kq * G::PLANE + y * G::PW + x;
    }
    const int aoff = G_IC * G::PLANE + kq * 9 * 16 + xl;             // wl[(ks*4 + kq)*9 + tap][oc = xl]
    f32x4 acc[G::NTW];
#pragma unroll
    for (int t = 0; t < G::NTW; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int my_tiles = (G::NTILES - wave + 3) >> 2;               // tiles wave, wave+4, ... < NTILES

    for (int c = 0; c < nchunks; ++c) {
        const float* buf = sm + (c & 1) * G::BUF;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const float a = buf[aoff + (ks * 4 * 9 + tap) * 16];
#pragma unroll
                for (int t = 0; t < G::NTW; ++t) {
                    if (t < G::NTW - 1 || t < my_tiles) {             // every wave owns at least NTW - 1 tiles
                        const float b = buf[boff[t] + ks * 4 * G::PLANE + (tap / 3) * G::PW + (tap % 3)];
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
                    }
                }
            }
        if (c + 1 < nchunks) {
            stash(sm + ((c + 1) & 1) * G::BUF);                      // the other buffer: its readers passed a barrier
            if (c + 2 < nchunks) fetch(c + 2);
        }
        __syncthreads();
    }

    {
        const int tiles = 2 * tiles_per_tower;
        const int tile = tower * tiles_per_tower + (oc0 >> 4);
        conv_tile_tail<HO>(acc, boff, my_tiles, hwv, P, tower, oc0, cpg, eps, sm, sm + 2 * G::BUF,
                           part + ((size_t)n * tiles + tile) * 4 * G::HW, tid, lane, wave, kq, xl);
    }
}

// GroupNorm + ReLU + partial heads for a convolution output that is already in memory: conv [N][2C][HO*HO], written by
// the blocked Winograd kernel (tower_wino.hip, BHO = HO).  workgroup = (track, tower, 16-channel tile) as above; the
// tile's 16 planes are re-loaded in the accumulator layout of the direct kernel and go through the same tail.  The four
// partial head planes of the tile are written over the tile's OWN first four convolution planes (nobody else reads
// them; this workgroup has them in registers): part[(n*tiles + tile) * 16 * HW + o * HW + p], tile stride 16 * HW.
template <int HO>
__global__ void __launch_bounds__(256, 2)
tower_gn_heads_kernel(float* __restrict__ conv, TowerParams P, int N, int C, int cpg, float eps, long long* trace) {
    using G = ConvGeom<HO>;
    extern __shared__ __attribute__((aligned(16))) float sm[];       // [16][PLANE] + [128]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __builtin_assume(wave >= 0 && wave < 4);                         // (lets the tail's position masks fold for all but the last tile)
    const int kq = lane >> 4, xl = lane & 15;
    const int tiles_per_tower = C >> 4;
    // consecutive workgroup ids go round-robin over the 8 XCDs: a track's workgroups run on the XCD whose L2 the blocked
    // Winograd kernel left the track's convolution output in (same arithmetic as there)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int n = (slot / (2 * tiles_per_tower)) * 8 + xcd;
    if (n >= N) return;
    // [grid][8] behind 65,536 workgroups' worth of slots (the other kernels of a head launch stamp the buffer's front):
    // start, sums, squares, planes, heads, end
    long long* tr = trace ? trace + ((size_t)65536 + blockIdx.x) * 8 : nullptr;
    if (tr && tid == 0) tr[0] = (long long)__builtin_amdgcn_s_memtime();
    const int rem = slot % (2 * tiles_per_tower);
    const int tower = rem / tiles_per_tower;
    const int oc0 = (rem - tower * tiles_per_tower) * 16;
    float hwv[9];
    {
        const int o = lane & 3, blk = lane >> 2;
        const float* wsrc = (tower == 1) ? P.reg_w + (size_t)o * C * 9
                                         : (o < 2 ? P.cls_w + (size_t)o * C * 9 : P.center_w);
        const bool live = (tower == 1) || (o < 3);
        wsrc += (size_t)oc0 * 9 + blk;
#pragma unroll
        for (int r = 0; r < 9; ++r) hwv[r] = live ? wsrc[r * 16] : 0.0f;
    }
    const int my_tiles = (G::NTILES - wave + 3) >> 2;
    float* __restrict__ base = conv + ((size_t)n * 2 * C + (size_t)tower * C + oc0) * G::HW;     // the tile's 16 planes
    int boff[G::NTW];
    f32x4 acc[G::NTW];
#pragma unroll
    for (int t = 0; t < G::NTW; ++t) {
        const int p = min(16 * (wave + 4 * t) + xl, G::HW - 1);
        const int y = p / HO, x = p - y * HO;
        boff[t] = kq * G::PLANE + y * G::PW + x;
        // unconditional loads at clamped positions (all in flight together); invalid columns are masked in the tail
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = base[(size_t)(4 * kq + r) * G::HW + p];
    }
    __builtin_amdgcn_sched_barrier(0);
    // the planes' halo while the loads are in flight (round 4's phase trace: the planes phase was 11.4 k of a workgroup's
    // 37.9 k cycles, the halo loop ~2.5 k of them on the critical path); the tail's barriers order it before the heads read
    conv_zero_halo<HO>(sm, tid);
    __builtin_amdgcn_sched_barrier(0);
    conv_tile_tail<HO, true>(acc, boff, my_tiles, hwv, P, tower, oc0, cpg, eps, sm, sm + 16 * G::PLANE, base, tid, lane, wave,
                             kq, xl, tr);
}

// logits[n][ch][pos] = bias[ch] + sum over the tower's tiles of the partial head sums (fixed tile order), ReLU on the
// four reg channels.  grid (N, 7, ceil(HW / 256)).
__global__ void __launch_bounds__(256)
heads_combine_hw_kernel(const float* __restrict__ part, int tpt, int HW, int planes_per_tile,
                        const float* __restrict__ cls_b, const float* __restrict__ center_b,
                        const float* __restrict__ reg_b, float* __restrict__ logits) {
    // planes_per_tile: 4 (the tile's four partial head planes back to back) or 16 (they sit at the head of the tile's
    // sixteen convolution planes: tower_gn_heads_kernel)
    const int n = blockIdx.x, ch = blockIdx.y;
    const int pos = blockIdx.z * 256 + threadIdx.x;
    if (pos >= HW) return;
    const int side = ch >= 3;
    const int o = side ? ch - 3 : ch;
    const float* __restrict__ p = part + (((size_t)n * 2 * tpt + side * tpt) * planes_per_tile + o) * HW + pos;
    float s = 0.0f;
    for (int t0 = 0; t0 < tpt; t0 += 8) {
        float v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = (t0 + t < tpt) ? p[(size_t)(t0 + t) * planes_per_tile * HW] : 0.0f;    // loads in flight together
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (t0 + t < tpt) s += v[t];
    }
    s += (ch < 2) ? cls_b[ch] : ((ch == 2) ? center_b[0] : reg_b[ch - 3]);
    if (side) s = relu_nan(s);
    logits[((size_t)n * 7 + ch) * HW + pos] = s;
}

// Towers + heads of a HOxHO response on the matrix cores: logits [N,7,HO,HO] complete on return (tower_ws holds the
// partial head sums in between).  SMOT_ERR_UNSUPPORTED when no instantiation fits (the caller then runs
// tower_generic_kernel + heads_kernel).
int launch_tower_conv(const float* resp, const TowerParams& P, int N, int C, int Ho, int cpg, float eps,
                      const float* cls_b, const float* center_b, const float* reg_b, float* tower_ws, float* logits,
                      unsigned* zero_words, hipStream_t st) {
    if (Ho != 29 || C % 16 != 0 || cpg > 16 || 16 % cpg != 0) return SMOT_ERR_UNSUPPORTED;
    using G = ConvGeom<29>;
    const size_t smem = (size_t)G::SMEM_FLOATS * sizeof(float);
    const int rco = ensure_lds_optin(reinterpret_cast<const void*>(&tower_conv_mfma_kernel<29>), smem,
                                     "predictor towers (conv, Ho=29)");
    if (rco) return rco;
    hipLaunchKernelGGL(tower_conv_mfma_kernel<29>, dim3(N * 2 * (C / 16)), dim3(256), smem, st, resp, P, C, cpg, eps,
                       tower_ws, zero_words);
    int rc = check_launch("predictor towers (conv, Ho=29)");
    if (rc) return rc;
    hipLaunchKernelGGL(heads_combine_hw_kernel, dim3(N, 7, (G::HW + 255) / 256), dim3(256), 0, st,
                       (const float*)tower_ws, C / 16, G::HW, 4, cls_b, center_b, reg_b, logits);
    return check_launch("predictor heads combine (Ho=29)");
}

// The same result through the blocked Winograd kernel (tower_wino.hip): convolution output of every 16x16 block of the
// 29x29 map -> tower_ws [N][2C][HW], then GroupNorm + ReLU + partial heads per (track, tower, 16-channel tile) in place,
// then the combine.  Needs the packed (transformed) filters of smot_emm_tower_pack.
int launch_tower_wino_blocks(const float* resp, const float* packed, const TowerParams& P, int N, int C, int cpg,
                             float* conv, unsigned* zero_words, hipStream_t st, const float* plane_max);   // tower_wino.hip
int launch_tower_conv_wino(const float* resp, const float* packed, const TowerParams& P, int N, int C, int Ho, int cpg,
                           float eps, const float* cls_b, const float* center_b, const float* reg_b, float* tower_ws,
                           float* logits, unsigned* zero_words, hipStream_t st, const float* plane_max) {
    if (Ho != 29 || C % 32 != 0 || cpg > 16 || 16 % cpg != 0 || packed == nullptr) return SMOT_ERR_UNSUPPORTED;
    using G = ConvGeom<29>;
    int rc = launch_tower_wino_blocks(resp, packed, P, N, C, cpg, tower_ws, zero_words, st, plane_max);
    if (rc) return rc;
    const size_t smem = (size_t)(16 * G::PLANE + 128) * sizeof(float);
    const int rco = ensure_lds_optin(reinterpret_cast<const void*>(&tower_gn_heads_kernel<29>), smem,
                                     "predictor GroupNorm + heads (Ho=29)");
    if (rco) return rco;
    hipLaunchKernelGGL(tower_gn_heads_kernel<29>, dim3(((N + 7) / 8) * 8 * 2 * (C / 16)), dim3(256), smem, st, tower_ws, P, N, C,
                       cpg, eps, g_trace);
    rc = check_launch("predictor GroupNorm + heads (Ho=29)");
    if (rc) return rc;
    hipLaunchKernelGGL(heads_combine_hw_kernel, dim3(N, 7, (G::HW + 255) / 256), dim3(256), 0, st,
                       (const float*)tower_ws, C / 16, G::HW, 16, cls_b, center_b, reg_b, logits);
    return check_launch("predictor heads combine (Ho=29)");
}

}  // namespace smot
