// wino_bf16_exp.h -- EXPERIMENT ONLY (-DBCNN_HIP_EXPERIMENT: libbcnn_hip_exp.so; never compiled into the product library):
// the fused F(2x2,3x3) forward / dX algorithm on the bf16 matrix pipe with split fp32 operands.
// Included by conv_winograd_fused.hip, whose argument structs, LDS constants and helpers it uses; kept with its parity
// test (tests/test_winograd.py: split_bf16_*) as the measured answer to "what would the bf16 pipe buy".
#pragma once

namespace bcnn_hip {

// Output transform + stores of one unit from the accumulators (wave (wrow, fh) holds row wrow of the 4 x 4 position matrix
// for 32 of the 64 channels): the column half on the registers, S[4][2][32][64] through LDS in two halves of 32
// channels, then one (channel, tile) per lane for the row half, bias / activation, masked raw buffer stores and the
// batch-norm statistics -- the epilogue of wino_fused_kernel as a function (used by the split-bf16 kernels).
template <int EPI, bool STATS>
__device__ __forceinline__ void wino_store_unit(const WinoFusedArgs& a, f32x16 (&acc)[4][2], float* S, rsrc_i4 rs_dst, rsrc_i4 rs_stats,
                                                int wid, int lane, bool e_whole, bool tile_ok, unsigned n, int th, int tw,
                                                int e_m0, int e_tb, int e_half) {
    const int l31 = lane & 31, wrow = wid >> 1, fh = wid & 1, HW = a.H * a.W;
    const bool odd_w = (a.W & 1) != 0;
    const int oh = 2 * th, ow = 2 * tw;
    const bool two_cols = ow + 1 < a.W, two_rows = oh + 1 < a.H;
    const unsigned o00 = tile_ok ? (n * (unsigned)a.M * (unsigned)HW + (unsigned)(oh * a.W + ow)) * 4u : kOOB;
    const unsigned o10 = (tile_ok && two_rows) ? o00 + (unsigned)a.W * 4u : kOOB;
    const unsigned o01 = two_cols ? o00 + 4u : kOOB, o11 = two_cols ? o10 + 4u : kOOB;
    const float w00 = tile_ok ? 1.f : 0.f, w10 = (tile_ok && two_rows) ? 1.f : 0.f, w01 = two_cols ? 1.f : 0.f;
    const unsigned st_voff = lane == 63 ? 0u : kOOB, st_voff2 = (lane == 63 && e_whole) ? 8u : kOOB;
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
        if (ph == 1) lds_barrier();
        if (fh == ph) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                if (tt == 1 && !e_whole) break;
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const buf_f32x2 m0v = {acc[0][tt][r], acc[0][tt][r + 1]}, m1v = {acc[1][tt][r], acc[1][tt][r + 1]};
                    const buf_f32x2 m2v = {acc[2][tt][r], acc[2][tt][r + 1]}, m3v = {acc[3][tt][r], acc[3][tt][r + 1]};
                    const buf_f32x2 s0 = m0v + m1v + m2v, s1 = m1v - m2v - m3v;
                    float* p = S + ((wrow * 2) * 32 + mfma_row(r, lane)) * WF_BT + tt * 32 + l31;
                    p[0] = s0[0];
                    p[WF_BT] = s0[1];
                    p[32 * WF_BT] = s1[0];
                    p[33 * WF_BT] = s1[1];
                }
            }
        }
        lds_barrier();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int fl = q * 8 + wid;
            const int f = e_m0 + ph * 32 + fl;
            const bool f_ok = f < a.M;
            float sb[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int b = 0; b < 2; ++b) sb[i][b] = S[((i * 2 + b) * 32 + fl) * WF_BT + lane];
            float o[2][2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                o[0][b] = sb[0][b] + sb[1][b] + sb[2][b];
                o[1][b] = sb[1][b] - sb[2][b] - sb[3][b];
            }
            if (EPI != 0) {
                float bv = (a.add_bias && f_ok) ? a.bias[f] : 0.f;
                if (bv == 1.0f) bv = 0.f;  // quirk 2
                const float sl = (EPI == 2 && a.act == BCNN_HIP_ACT_PRELU && a.slopes && f_ok) ? a.slopes[f] : 0.f;
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        float vv = o[r][c];
                        if (bv != 0.0f) vv += bv;
                        if (EPI == 1) vv = vv * (float)(vv > 0);
                        else if (a.act != BCNN_HIP_ACT_NONE) vv = act_fwd_cheap(vv, a.act, sl);
                        o[r][c] = vv;
                    }
            }
            const int soff = f_ok ? f * HW * 4 : 0;
            const unsigned kill = f_ok ? 0u : kOOB;
            if (!odd_w) {
                buffer_store_f32x2(buf_f32x2{o[0][0], o[0][1]}, rs_dst, (int)(o00 | kill), soff, 0);
                buffer_store_f32x2(buf_f32x2{o[1][0], o[1][1]}, rs_dst, (int)(o10 | kill), soff, 0);
            } else {
                buffer_store_f32(o[0][0], rs_dst, (int)(o00 | kill), soff, 0);
                buffer_store_f32(o[0][1], rs_dst, (int)(o01 | kill), soff, 0);
                buffer_store_f32(o[1][0], rs_dst, (int)(o10 | kill), soff, 0);
                buffer_store_f32(o[1][1], rs_dst, (int)(o11 | kill), soff, 0);
            }
            if (STATS) {
                const float sv0 = (o[0][0] + o[0][1] * w01) * w00, sv1 = (o[1][0] + o[1][1] * w01) * w10;
                const float sq0 = (o[0][0] * o[0][0] + o[0][1] * o[0][1] * w01) * w00;
                const float sq1 = (o[1][0] * o[1][0] + o[1][1] * o[1][1] * w01) * w10;
                const float sv = wave_sum_dpp(sv0 + sv1), sq = wave_sum_dpp(sq0 + sq1);
                const int st_soff = f_ok ? (f * (2 * a.tblocks) + 2 * e_tb + e_half) * 8 : 0;
                buffer_store_f32x2(buf_f32x2{sv, sq}, rs_stats, (int)(st_voff | kill), st_soff, 0);
                buffer_store_f32x2(buf_f32x2{0.f, 0.f}, rs_stats, (int)(st_voff2 | kill), st_soff, 0);
            }
        }
    }
}

// =============================================================================================================
// The same fused forward / dX algorithm on the bf16 matrix pipe with fp32-equivalent accuracy ("split" arithmetic):
// every fp32 operand is the sum of NP bf16 parts (round to nearest: x = x0 + x1 [+ x2], |x - sum| <= 2^-17 |x| for two
// parts, 2^-25 for three) and a product is the sum of the part products that matter -- x0*y0 + x0*y1 + x1*y0 for NP = 2,
// plus x0*y2 + x2*y0 + x1*y1 for NP = 3 -- accumulated in fp32 by v_mfma_f32_32x32x16_bf16. One such MFMA covers 16
// reduction channels in ~32 cycles where v_mfma_f32_32x32x2_f32 needs 8 x 64 (tools/micro/mfma_bf16.hip), so even six
// part products cost 2.7x less matrix-pipe time than exact fp32. Measured error against float64 (tools/exp/bf16_split.py,
// ResNet shapes): NP = 3: 1e-7 (the fp32 Winograd kernel above: 3e-7 .. 1e-6); NP = 2: 7e-6 .. 9e-6.
//
// Layout of the work is that of wino_fused_kernel (64 tiles x 64 channels x 16 positions per unit, wave (wrow, fh)),
// but a chunk is 16 input channels and has two phases separated by barriers (V is single-buffered: the vector ALU and
// the matrix pipe do not overlap on a SIMD anyway, so nothing is lost by not overlapping them across waves):
//   A  wave w transforms channels 2w, 2w+1 of the chunk for the 64 tiles, splits the 16 V values of both into bf16
//      parts and stores them as (k, k+1) pairs -- one 32-bit word per tile: V[part][xi][pair w][tile];
//      its own U operands of the chunk (4 positions x NP parts x 16 bytes per lane, packed by wino_pack_bf16_kernel in
//      MFMA operand order) are requested from global memory at the start of the phase, the patches of the NEXT chunk at
//      its end;
//   B  4 positions x 2 tile halves x (3 or 6) MFMAs per wave, B operands from LDS.
// =============================================================================================================
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
constexpr int WB_KC = 16;                       // channels per chunk = K of the MFMA
#ifndef WB_DEFAULT_PARTS
#define WB_DEFAULT_PARTS 0  // 0: exact-fp32 kernel; 2 / 3: split-bf16 kernel with that many parts
#endif
constexpr int WB_VPART = 16 * (WB_KC / 2) * 64;  // 32-bit words per V part

template <int EPI, bool STATS, int NP>
__global__ __launch_bounds__(512, 2) void wino_bf16_kernel(const WinoFusedArgs a) {
    // V parts (NP x 32 KB); the epilogue's S (64 KB) reuses the space
    __shared__ __attribute__((aligned(16))) unsigned ldsw[(NP * WB_VPART > WF_STAGE) ? NP * WB_VPART : WF_STAGE];
    float* const lds = reinterpret_cast<float*>(ldsw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int HW = a.H * a.W;
    const int wrow = wid >> 1, fh = wid & 1;
    const rsrc_i4 rs_src = make_rsrc(a.src, a.src_bytes);
    const rsrc_i4 rs_u = make_rsrc(a.upk, a.upk_bytes);
    const rsrc_i4 rs_dst = make_rsrc(a.dst, a.dst_bytes);
    const rsrc_i4 rs_stats = make_rsrc(a.stats, STATS ? a.stats_bytes : 0u);
    const unsigned per_img = (unsigned)(a.TH * a.TW);
    const int nchunks = a.Jpad / WB_KC;
    const bool odd_w = (a.W & 1) != 0;

    // per-unit state: of the unit being multiplied and, from the end of its K loop on, of the NEXT unit (whose first
    // patches are requested before the epilogue of the current one)
    bool whole = false, tile_ok = false, pad_l = false, pad_r = false;
    int m0 = 0, tb = 0, half = 0, th = 0, tw = 0;
    unsigned n = 0;
    unsigned voff[4], voff_edge[4];
    auto decode = [&](int unit) {
        whole = unit < a.nfull;
        const int blk = whole ? unit : a.nfull + ((unit - a.nfull) >> 1);
        half = whole ? 0 : ((unit - a.nfull) & 1);
        const int mb = blk % a.mblocks;
        tb = blk / a.mblocks;
        m0 = mb * WF_BF;
        const unsigned t = (unsigned)tb * WF_BT + (unsigned)(half * 32) + (unsigned)lane;
        tile_ok = t < a.T && (whole || lane < 32);
        const bool addr_ok = t < a.T;  // lanes 32-63 of a half block load their (real) tiles too (neighbour columns)
        n = addr_ok ? t / per_img : 0u;
        const unsigned rr = addr_ok ? t - n * per_img : 0u;
        th = (int)(rr / (unsigned)a.TW);
        tw = (int)(rr - (unsigned)th * (unsigned)a.TW);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ih = 2 * th - 1 + i;
            const bool row_ok = addr_ok && (unsigned)ih < (unsigned)a.H;
            const unsigned row = ((n * (unsigned)a.J * (unsigned)HW) + (unsigned)(ih * a.W)) * 4u;
            voff[i] = row_ok ? row + (unsigned)(8 * tw) : kOOB;
            const bool want_l = lane == 0 && tw > 0, want_r = lane == 63 && tw + 1 < a.TW;
            voff_edge[i] = (row_ok && (want_l || want_r)) ? row + (unsigned)(want_l ? 8 * tw - 4 : 8 * tw + 8) : kOOB;
#ifdef WB_ABL_LOADS_OOB  // timing experiment: no patch load touches memory
            voff[i] = kOOB; voff_edge[i] = kOOB;
#endif
        }
        pad_l = tw == 0;
        pad_r = tw + 1 == a.TW;
    };
    // raw patches of this wave's two channels: per row the pair (columns 2tw, 2tw+1) and the neighbour column that
    // lanes 0 / 63 fetch themselves
    buf_f32x2 pm[2][4];
    float pe[2][4];
    auto load_patches = [&](int kc, unsigned kill) {  // kill = kOOB: request nothing (past the last chunk)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const unsigned soff = (unsigned)(kc * WB_KC + 2 * wid + c) * (unsigned)HW * 4u;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                pm[c][i] = buffer_load_f32x2(rs_src, (int)(voff[i] | kill), (int)soff, 0);
                pe[c][i] = buffer_load_f32(rs_src, (int)(voff_edge[i] | kill), (int)soff, 0);
            }
        }
    };
    auto transform = [&](int c, float (&v)[16]) {  // B^T d B of channel c of the pair
        float d[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = __builtin_bit_cast(int, pe[c][i]);
            float m1 = pm[c][i][1];
            if (odd_w) m1 = pad_r ? 0.f : m1;  // uniform
            const int l = __builtin_amdgcn_update_dpp(e, __builtin_bit_cast(int, m1), 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
            const int r = __builtin_amdgcn_update_dpp(e, __builtin_bit_cast(int, pm[c][i][0]), 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
            d[i][0] = pad_l ? 0.f : __builtin_bit_cast(float, l);
            d[i][1] = pm[c][i][0];
            d[i][2] = m1;
            d[i][3] = pad_r ? 0.f : __builtin_bit_cast(float, r);
        }
        float tt[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tt[0][j] = d[0][j] - d[2][j];
            tt[1][j] = d[1][j] + d[2][j];
            tt[2][j] = d[2][j] - d[1][j];
            tt[3][j] = d[1][j] - d[3][j];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[4 * i + 0] = tt[i][0] - tt[i][2];
            v[4 * i + 1] = tt[i][1] + tt[i][2];
            v[4 * i + 2] = tt[i][2] - tt[i][1];
            v[4 * i + 3] = tt[i][1] - tt[i][3];
        }
    };

    decode(blockIdx.x);
    load_patches(0, 0u);
    for (int unit = blockIdx.x; unit < a.nunits; unit += gridDim.x) {
#ifdef WF_ABL_CLOCK
        const bool stamp_on = blockIdx.x == 37 && unit == (int)blockIdx.x + (int)gridDim.x;
#endif
        WF_STAMP(0);
        f32x16 acc[4][2];  // not cleared: the first MFMA of chunk 0 multiplies onto a literal zero
        // U operands of this wave: lane (channel fh*32 + l31, k group lhi) reads 8 consecutive k of one part: 16 bytes
        const unsigned u_voff = ((unsigned)lhi * (unsigned)a.Mpad + (unsigned)(m0 + fh * 32 + l31)) * 16u;
        buf_f32x4 ua[4][NP];

        load_patches(0, 0u);
        for (int kc = 0; kc < nchunks; ++kc) {
            // ---- phase A ---------------------------------------------------------------------------------
            if (kc < 6) WF_STAMP(4 + 4 * kc);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const unsigned soff = ((unsigned)((p * 16 + 4 * wrow + j) * nchunks + kc) * 2u * (unsigned)a.Mpad) * 16u;
                    ua[j][p] = buffer_load_f32x4(rs_u, (int)u_voff, (int)soff, 0);
                }
            {
                float v0[16], v1[16];
                transform(0, v0);
                transform(1, v1);
                if (NP == 2) load_patches(kc + 1, kc + 1 < nchunks ? 0u : kOOB);  // the registers are free again
                unsigned* vw = ldsw + wid * 64 + lane;
#pragma unroll
                for (int xi = 0; xi < 16; ++xi) {
                    float r0 = v0[xi], r1 = v1[xi];
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(buf_f32x2{r0, r1}, bf16x2_t));
                        vw[(p * 16 + xi) * (WB_KC / 2) * 64] = h;
                        if (p + 1 < NP) {  // exact remainders
                            r0 -= __builtin_bit_cast(float, h << 16);
                            r1 -= __builtin_bit_cast(float, h & 0xffff0000u);
                        }
                    }
                }
            }
            if (NP != 2) load_patches(kc + 1, kc + 1 < nchunks ? 0u : kOOB);  // (three parts: too few registers to go earlier)
            if (kc < 6) WF_STAMP(5 + 4 * kc);
            lds_barrier();
            if (kc < 6) WF_STAMP(6 + 4 * kc);
            // ---- phase B ---------------------------------------------------------------------------------
            const unsigned* vr = ldsw + (4 * lhi) * 64 + l31;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int xi = 4 * wrow + j;
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    if (tt == 1 && !whole) break;  // uniform
                    bf16x8_t vb[NP];
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        const unsigned* q = vr + (p * 16 + xi) * (WB_KC / 2) * 64 + tt * 32;
                        const unsigned w4[4] = {q[0], q[64], q[128], q[192]};
                        vb[p] = __builtin_bit_cast(bf16x8_t, w4);
                    }
                    f32x16 c = acc[j][tt];
                    // smallest terms first
                    if (NP == 3) {
                        if (kc == 0) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ua[j][1]), vb[1], f32x16{0}, 0, 0, 0);
                        else c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ua[j][1]), vb[1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ua[j][0]), vb[2], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ua[j][2]), vb[0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ua[j][0]), vb[1], c, 0, 0, 0);
                    } else {
                        if (kc == 0) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ua[j][0]), vb[1], f32x16{0}, 0, 0, 0);
                        else c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ua[j][0]), vb[1], c, 0, 0, 0);
                    }
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ua[j][1]), vb[0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ua[j][0]), vb[0], c, 0, 0, 0);
                    acc[j][tt] = c;
                }
            }
            if (kc < 6) WF_STAMP(7 + 4 * kc);
            lds_barrier();  // every wave is done with V (the next phase A, or the epilogue's S, overwrites it)
        }

        // ---- epilogue (S in the V space); the next unit's first patches fly under it ------------------------------
        WF_STAMP(28);
        {
            const bool e_whole = whole, e_tile_ok = tile_ok;
            const int e_m0 = m0, e_tb = tb, e_half = half, e_th = th, e_tw = tw;
            const unsigned e_n = n;
            if (unit + (int)gridDim.x < a.nunits) {
                decode(unit + (int)gridDim.x);
                load_patches(0, 0u);
            }
            wino_store_unit<EPI, STATS>(a, acc, lds, rs_dst, rs_stats, wid, lane, e_whole, e_tile_ok, e_n, e_th, e_tw, e_m0, e_tb, e_half);
        }
        WF_STAMP(31);
        lds_barrier();  // S has been read: the next unit's V may be written
    }
}

// U = G g G^T split into NP bf16 parts in MFMA A-operand order: [part][xi][chunk of 16 j][j / 8 % 2][m][j % 8]
template <int NP>
__global__ __launch_bounds__(256) void wino_pack_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ u, int F, int C,
                                                             int dx_mode, int Jpad, int Mpad) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Jpad * Mpad) return;
    const int j = idx / Mpad, m = idx - j * Mpad;
    const int M = dx_mode ? C : F, J = dx_mode ? F : C;
    float t[4][3];
    if (m < M && j < J) {
        const int f = dx_mode ? j : m, c = dx_mode ? m : j;
        const float* p = w + ((size_t)f * C + c) * 9;
        float g[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int b = 0; b < 3; ++b) g[r][b] = dx_mode ? p[(2 - r) * 3 + (2 - b)] : p[r * 3 + b];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            t[0][b] = g[0][b];
            t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
            t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
            t[3][b] = g[2][b];
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int b = 0; b < 3; ++b) t[r][b] = 0.f;
    }
    const int nch = Jpad / WB_KC;
    const size_t base = ((size_t)((j / WB_KC) * 2 + ((j >> 3) & 1)) * Mpad + m) * 8 + (j & 7);
    const size_t xi_stride = (size_t)nch * 2 * Mpad * 8, part_stride = 16 * xi_stride;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float uv[4] = {t[r][0], 0.5f * (t[r][0] + t[r][1] + t[r][2]), 0.5f * (t[r][0] - t[r][1] + t[r][2]), t[r][2]};
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            float rem = uv[b];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const __bf16 h = (__bf16)rem;  // round to nearest even
                u[p * part_stride + (size_t)(4 * r + b) * xi_stride + base] = h;
                rem -= (float)h;
            }
        }
    }
}


// ---- host side ----
// Split-bf16 form (wino_bf16_kernel): which layers and how many parts. BCNN_HIP_WINOGRAD_BF16 = 0 / 2 / 3 forces it in the
// experiment build.
static int g_wb_parts = -1;
static int wino_bf16_parts(int J) {
    if (J % WB_KC != 0) return 0;
    if (g_wb_parts < 0) {
        const char* e = BCNN_EXP_ENV("BCNN_HIP_WINOGRAD_BF16");
        g_wb_parts = e ? (e[0] == '0' ? 0 : (e[0] == '2' ? 2 : 3)) : WB_DEFAULT_PARTS;
    }
    return g_wb_parts;
}

template <int NP>
static void wino_bf16_launch(WinoFusedArgs& a, const float* w, const ConvShape& s, int dx_mode, unsigned grid, bool plain) {
    const size_t u_elems = (size_t)NP * 16 * a.Jpad * a.Mpad;  // bf16
    a.upk_bytes = (unsigned)(u_elems * 2);
    float* U = wf_scratch((u_elems + 1) / 2);
    a.upk = U;
    wino_pack_bf16_kernel<NP><<<ceil_div((long long)a.Jpad * a.Mpad, 256), 256, 0, current_stream()>>>(
        w, reinterpret_cast<__bf16*>(U), s.F, s.C, dx_mode, a.Jpad, a.Mpad);
    KERNEL_CHECK();
    if (a.stats) wino_bf16_kernel<0, true, NP><<<grid, 512, 0, current_stream()>>>(a);
    else if (plain) wino_bf16_kernel<0, false, NP><<<grid, 512, 0, current_stream()>>>(a);
    else if (a.act == BCNN_HIP_ACT_RELU) wino_bf16_kernel<1, false, NP><<<grid, 512, 0, current_stream()>>>(a);
    else wino_bf16_kernel<2, false, NP><<<grid, 512, 0, current_stream()>>>(a);
    KERNEL_CHECK();
}


}  // namespace bcnn_hip
