// conv_winograd_fused.hip -- Winograd F(2x2, 3x3) forward / dX in ONE kernel, for the wide-and-shallow 3x3 / s1
// layers (64 and 128 channels at 56x56 / 28x28) where the three-kernel form of conv_winograd.hip loses: there the
// transformed tensors V and M are 4x the activation each (0.4 GB per layer at N = 128) and streaming them through HBM
// costs more than the 2.25x fewer MACs save. Here V only ever exists in LDS and M only in the accumulators:
//
//   workgroup = 64 output channels x 64 tiles (2x2 outputs each) x all 16 Winograd positions xi, 8 waves;
//   wave w owns ROW i = w / 2 of the 4x4 position matrix (xi = 4i .. 4i+3) for 32 of the 64 channels:
//   acc[4 positions][2 t-tiles] of 32x32 (128 accumulator registers);
//   K loop over 8 input channels at a time:
//     U chunk [16][8][64 f]  transformed weights, global -> LDS by LDS-DMA (16 B per lane, 4 rows per instruction);
//     V chunk [16][8][64 t]  wave w loads the 4x4 patches of channel w for the 64 tiles straight into registers
//                            (buffer loads: an out-of-range offset returns the zero padding), requested BEFORE the
//                            MFMAs of the current chunk and transformed (B^T d B, 32 adds) and written to the other
//                            LDS stage AFTER them -- the loads fly under 32 MFMAs per wave;
//     32 v_mfma_f32_32x32x2_f32 per wave and chunk, operands by ds_read_b32 (conflict-free: f / t contiguous);
//   epilogue: a wave holds a whole row of M, so the column half of the output transform (M A: 4 -> 2 values) is done
//   on the accumulators; S[i][b][f][t] (128 KB) goes through LDS once, then one (f, t) per lane finishes A^T S,
//   adds bias / activation, stores 8 bytes per row, and for a fused batch-norm reduces the per-channel sum / sum
//   of squares of the stored values over the wave (a wave holds one channel x 64 tiles).
// dX of such a layer is the same convolution of dy with the rotated, transposed filter.
//
// Reference: bcnn_forward_conv_layer_cpu's Winograd branch (bcnn_conv_layer.c:388-436) on bcnn_mat.c:1403-2138
// (PREDICT mode there; here also TRAIN forward / dX, inside the 1e-4 parity bar).
#include "conv_common.h"
#include "lds_dma.h"
#include "wino43_pack.h"

namespace bcnn_hip {

constexpr int WF_BT = 64;  // tiles per workgroup
constexpr int WF_BF = 64;  // output channels per workgroup
constexpr int WF_KC = 8;   // reduction channels per chunk == waves per workgroup (wave w transforms channel w)
constexpr int WF_STAGE = 16 * WF_KC * (WF_BF + WF_BT);  // floats per LDS stage: U then V

struct WinoFusedArgs {
    const float* src;  // x (forward) or dy (dX): [N][J][H][W]
    const float* upk;  // transformed weights [16][Jpad][Mpad], zero padded
    float* dst;        // [N][M][H][W]
    const float* bias;
    const float* slopes;
    float* stats;      // optional: [M][2 * tblocks][2] (one slot per half block)
    int N, J, M, H, W, TH, TW;
    unsigned T;
    int Jpad, Mpad, mblocks, tblocks;
    int nfull;         // units [0, nfull) are whole 64-tile blocks, the rest are the two 32-tile halves of the last blocks
    int nunits;
    int act, add_bias;
    unsigned src_bytes, upk_bytes, dst_bytes, stats_bytes;
    // K-split tail (EPI == 0 kernels, tail_units > 0): units [nunits, nunits + tail_units) -- the blocks a last, partly
    // filled round would hold -- are not run as units. Their chunks, flattened (unit-major), are dealt out evenly:
    // workgroup i takes chunks [i * tail_q, (i + 1) * tail_q) as one or two PIECES (a piece = a chunk range of one unit) and
    // writes each piece's raw 2 x 2 outputs -- a partial sum over its input channels; the output transform is linear --
    // to tail_scr[2 * i + piece][channel 0..63][tile 0..63][4]. wino_tail_fixup_kernel adds a unit's pieces in channel
    // order and stores / takes the statistics like the epilogue here does.
    float* tail_scr;
    int tail_q, tail_units;
    unsigned tail_scr_bytes;
};

#ifndef WF_LATE_AFTER
#define WF_LATE_AFTER 1  // the late half produces after this k-step (0..3) of its multiply
#endif

#define WF_STAMP(i) do { } while (0)

// EPI: 0 = plain store, 1 = bias + ReLU, 2 = bias + any other cheap activation (runtime switch); STATS: per-channel
// sum / sum of squares of the raw outputs for a batch-norm that follows (EPI == 0 only)
//
// Persistent: one workgroup per CU (128 KB of LDS) walks the (tile block, channel block) pairs with stride gridDim.x,
// so no CU waits for a dispatch between blocks. Per chunk every wave "produces" once (transforms the NEXT chunk's
// patches into V, requests the patches of the chunk after that, DMAs its share of the next U slab) and multiplies
// (32 MFMAs). Vector-ALU work and fp32 MFMAs do not overlap on a SIMD, but each SIMD hosts one
// wave of each half of the workgroup and the halves produce at different times -- waves 0-3 before their MFMAs,
// waves 4-7 between their second and third k-step -- so that whenever one wave stalls in its produce step (memory
// latency, LDS stores) the other one has MFMAs to issue (cycle stamps: tools/exp/wf_clock.py).
//
// Patch loads: a patch row is the pair of columns (2 tw, 2 tw + 1) -- ONE 8-byte load per lane, contiguous over the
// wave's tiles -- plus its left / right neighbour columns, which are the neighbouring lanes' pairs (wave shifts; only
// lanes 0 and 63 fetch theirs from memory). 8 load instructions and ~1/4 of the L1 line traffic of 16 stride-2 dword
// loads, whose issue was measured to stall the requesting waves ~1250 cycles per chunk. (Raw buffer loads of 8 bytes
// need 4-byte alignment only and are range-checked per dword: tools/micro/bufload_probe.hip.)
template <int EPI, bool STATS>
__global__ __launch_bounds__(512, 2) void wino_fused_kernel(const WinoFusedArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2 * WF_STAGE];  // 128 KB: two stages; the epilogue's S in the second one
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int HW = a.H * a.W;
    const int wrow = wid >> 1, fh = wid & 1;  // this wave's row of positions and its half of the 64 channels
    const bool early = wid < 4;               // transforms at the head of a chunk (see above)
    const rsrc_i4 rs_src = make_rsrc(a.src, a.src_bytes);
    const rsrc_i4 rs_u = make_rsrc(a.upk, a.upk_bytes);
    const rsrc_i4 rs_dst = make_rsrc(a.dst, a.dst_bytes);
    const rsrc_i4 rs_stats = make_rsrc(a.stats, STATS ? a.stats_bytes : 0u);
    const rsrc_i4 rs_scr = make_rsrc(a.tail_scr, EPI == 0 ? a.tail_scr_bytes : 0u);
    const unsigned lds0 = lds_offset(&lds[0]);
    // LDS-DMA of U: 4 rows (k) x 64 floats per instruction; lane -> row lane / 16, floats 4 * (lane % 16) ..
    const unsigned u_voff = ((unsigned)(lane >> 4) * (unsigned)a.Mpad + (unsigned)(lane & 15) * 4u) * 4u;
    const unsigned per_img = (unsigned)(a.TH * a.TW);
    const int nchunks_full = a.Jpad / WF_KC;
    // the piece being multiplied (and, from the end of its K loop on, the NEXT one): a whole unit is the piece
    // [0, nchunks_full) with slot < 0; kb = its first chunk, nchunks = how many
    int kb = 0, nchunks = nchunks_full, slot = -1;

    // ---- per-unit state: of the unit being multiplied and, from the end of its K loop on, of the NEXT unit ----------
    // Work units: whole blocks first; when the last round would leave most CUs idle, its blocks are split into
    // two 32-tile halves (second t-tile of the accumulators unused) so that twice as many CUs share that round.
    bool whole = false, tile_ok = false, pad_l = false, pad_r = false;
    int m0 = 0, tb = 0, half = 0, th = 0, tw = 0;
    unsigned n = 0;
    unsigned voff[4], voff_edge[4];
    const bool odd_w = (a.W & 1) != 0;  // then the last tile's column 2 tw + 1 == W is padding as well
    auto decode = [&](int unit) {
        whole = unit < a.nfull;
        const int blk = whole ? unit : a.nfull + ((unit - a.nfull) >> 1);
        half = whole ? 0 : ((unit - a.nfull) & 1);
        const int mb = blk % a.mblocks;  // channel blocks of one tile block run together
        tb = blk / a.mblocks;
        m0 = mb * WF_BF;
        // this lane's tile (the same one for the input transform and for the output transform)
        const unsigned t = (unsigned)tb * WF_BT + (unsigned)(half * 32) + (unsigned)lane;
        tile_ok = t < a.T && (whole || lane < 32);
        // lanes 32-63 of a half block load their (real) tiles too: lane 31 takes its right column from lane 32
        const bool addr_ok = t < a.T;
        n = addr_ok ? t / per_img : 0u;
        const unsigned rr = addr_ok ? t - n * per_img : 0u;
        th = (int)(rr / (unsigned)a.TW);
        tw = (int)(rr - (unsigned)th * (unsigned)a.TW);
        const int ih0 = 2 * th - 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ih = ih0 + i;
            const bool row_ok = addr_ok && (unsigned)ih < (unsigned)a.H;
            const unsigned row = ((n * (unsigned)a.J * (unsigned)HW) + (unsigned)(ih * a.W)) * 4u;
            voff[i] = row_ok ? row + (unsigned)(8 * tw) : kOOB;
            const bool want_l = lane == 0 && tw > 0, want_r = lane == 63 && tw + 1 < a.TW;
            voff_edge[i] = (row_ok && (want_l || want_r)) ? row + (unsigned)(want_l ? 8 * tw - 4 : 8 * tw + 8) : kOOB;
        }
        pad_l = tw == 0;
        pad_r = tw + 1 == a.TW;
    };

    // d[i][1..2] = the pair, d[i][0] = the neighbour column fetched by lanes 0 / 63 (0.0 in all other lanes)
    float d[4][4];
    auto load_patch = [&](int kc) {  // channel kc*8 + wid of this lane's tile
        const unsigned soff = (unsigned)((kb + kc) * WF_KC + wid) * (unsigned)HW * 4u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const buf_f32x2 m = buffer_load_f32x2(rs_src, (int)voff[i], (int)soff, 0);
            d[i][1] = m[0]; d[i][2] = m[1];
            d[i][0] = buffer_load_f32(rs_src, (int)voff_edge[i], (int)soff, 0);
        }
    };
    auto dma_u = [&](int kc, int stage) {  // each wave brings in two positions: 2 x 8 rows of 64 floats
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int xi = 2 * wid + (q >> 1), r0 = (q & 1) * 4;
            const unsigned soff = (((unsigned)xi * (unsigned)a.Jpad + (unsigned)((kb + kc) * WF_KC + r0)) * (unsigned)a.Mpad + (unsigned)m0) * 4u;
            dma_row_x4(rs_u, lds0 + (unsigned)((stage * WF_STAGE + (xi * WF_KC + r0) * WF_BF) * 4), u_voff, soff);
        }
    };
    auto write_v = [&](int stage) {  // B^T d B -> V[xi][wid][lane]
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // lane l - 1's right element / lane l + 1's left element; lanes 0 / 63 keep `old`
            const int e = __builtin_bit_cast(int, d[i][0]);
            const int l = __builtin_amdgcn_update_dpp(e, __builtin_bit_cast(int, d[i][2]), 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
            const int r = __builtin_amdgcn_update_dpp(e, __builtin_bit_cast(int, d[i][1]), 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
            d[i][0] = pad_l ? 0.f : __builtin_bit_cast(float, l);
            d[i][3] = pad_r ? 0.f : __builtin_bit_cast(float, r);
            if (odd_w) d[i][2] = pad_r ? 0.f : d[i][2];  // uniform
        }
        float tt[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tt[0][j] = d[0][j] - d[2][j];
            tt[1][j] = d[1][j] + d[2][j];
            tt[2][j] = d[2][j] - d[1][j];
            tt[3][j] = d[1][j] - d[3][j];
        }
        float* v = lds + stage * WF_STAGE + 16 * WF_KC * WF_BF + wid * WF_BT + lane;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[(4 * i + 0) * WF_KC * WF_BT] = tt[i][0] - tt[i][2];
            v[(4 * i + 1) * WF_KC * WF_BT] = tt[i][1] + tt[i][2];
            v[(4 * i + 2) * WF_KC * WF_BT] = tt[i][2] - tt[i][1];
            v[(4 * i + 3) * WF_KC * WF_BT] = tt[i][1] - tt[i][3];
        }
    };
    // chunk 0 of a unit: U and V into stage 0, the patches of chunk 1 into the registers. The first unit's is issued
    // here, every later unit's during the previous unit's epilogue (which keeps its S in stage 1 only).
    // this workgroup's work list: its whole units, then (K-split tail) one or two pieces
    const int nreg = (int)blockIdx.x < a.nunits ? (a.nunits - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    int npieces = 0, pa_unit = 0, pa_k0 = 0, pa_n = 0, pb_n = 0;
    if (EPI == 0 && a.tail_units > 0) {
        const int c0 = (int)blockIdx.x * a.tail_q, c1 = min(c0 + a.tail_q, a.tail_units * nchunks_full);
        if (c0 < c1) {
            pa_unit = a.nunits + c0 / nchunks_full;
            pa_k0 = c0 % nchunks_full;
            pa_n = min(c1 - c0, nchunks_full - pa_k0);
            pb_n = (c1 - c0) - pa_n;  // > 0: the range runs on into the next unit
            npieces = pb_n > 0 ? 2 : 1;
        }
    }
    const int nitems = nreg + npieces;
    auto start_unit = [&](int it) {  // item `it` of the list becomes the piece being loaded
        int unit;
        if (it < nreg) { unit = (int)blockIdx.x + it * (int)gridDim.x; kb = 0; nchunks = nchunks_full; slot = -1; }
        else if (it == nreg) { unit = pa_unit; kb = pa_k0; nchunks = pa_n; slot = 2 * (int)blockIdx.x; }
        else { unit = pa_unit + 1; kb = 0; nchunks = pb_n; slot = 2 * (int)blockIdx.x + 1; }
        decode(unit);
        dma_u(0, 0);
        load_patch(0);
    };
    auto finish_start = [&]() {
        write_v(0);
        if (nchunks > 1) load_patch(1);  // every wave holds the next chunk's patches in registers
    };

    if (nitems == 0) return;
    start_unit(0);
    finish_start();
    for (int it = 0; it < nitems; ++it) {
        WF_STAMP(0);
        WF_STAMP(1);
        WF_STAMP(2);
        // U(0) is older than the (up to) 8 patch requests of chunk 1, which may fly on
        if (nchunks > 1) dma_wait_n<8>(); else dma_wait();
        __syncthreads();  // stage 0 holds chunk 0 of this unit; the previous unit's epilogue has read its S (stage 1)
        WF_STAMP(3);

        f32x16 acc[4][2];  // not cleared: the first k-step of chunk 0 multiplies onto a literal zero

        // everything chunk kc contributes to the chunks after it: V of chunk kc + 1 from the patches requested a whole
        // chunk ago, the patch requests of chunk kc + 2 into the registers this frees, the U slab of chunk kc + 1
        auto produce = [&](int kc) {
            if (kc + 1 < nchunks) write_v((kc & 1) ^ 1);
            // the U slab first: it is what the barrier at the end of this chunk waits for -- the patch requests behind it
            // (needed a whole chunk later) may still be in flight then (vmcnt retires in order)
            if (kc + 1 < nchunks) dma_u(kc + 1, (kc & 1) ^ 1);
            if (kc + 2 < nchunks) load_patch(kc + 2);
        };
        // 32 MFMAs on stage `cur`; the late half produces (see above) between the second and the third k-step
        auto multiply = [&](int cur, int produce_kc, bool first, int kc_stamp) {
            const float* us = lds + cur * WF_STAGE + (4 * wrow) * WF_KC * WF_BF + fh * 32 + l31;
            const float* vs = lds + cur * WF_STAGE + 16 * WF_KC * WF_BF + (4 * wrow) * WF_KC * WF_BT + l31;
            // fragments of k-step ks + 1 are fetched from LDS before the MFMAs of k-step ks are issued
            float af[2][4], bf[2][4][2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                af[0][j] = us[(j * WF_KC + lhi) * WF_BF];
                bf[0][j][0] = vs[(j * WF_KC + lhi) * WF_BT];
                bf[0][j][1] = vs[(j * WF_KC + lhi) * WF_BT + 32];
            }
#pragma unroll
            for (int ks = 0; ks < WF_KC / 2; ++ks) {
                const int fc = ks & 1, fn = fc ^ 1;
                if (ks + 1 < WF_KC / 2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        af[fn][j] = us[(j * WF_KC + 2 * ks + 2 + lhi) * WF_BF];
                        bf[fn][j][0] = vs[(j * WF_KC + 2 * ks + 2 + lhi) * WF_BT];
                        bf[fn][j][1] = vs[(j * WF_KC + 2 * ks + 2 + lhi) * WF_BT + 32];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (ks == 0 && first) {  // uniform
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[j][0] = mfma32(af[fc][j], bf[fc][j][0], zero);
                        if (whole) acc[j][1] = mfma32(af[fc][j], bf[fc][j][1], zero);  // uniform
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[j][0] = mfma32(af[fc][j], bf[fc][j][0], acc[j][0]);
                        if (whole) acc[j][1] = mfma32(af[fc][j], bf[fc][j][1], acc[j][1]);  // uniform
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (ks == WF_LATE_AFTER && produce_kc >= 0) {  // uniform
                    produce(produce_kc);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };

        // ---- K loop -------------------------------------------------------------------------------------
        for (int kc = 0; kc < nchunks; ++kc) {
            const int cur = kc & 1;
            if (kc < 8) WF_STAMP(4 + 3 * kc);
            // the other stage was last read before the barrier that ended iteration kc - 1
            if (early) produce(kc);
            __builtin_amdgcn_sched_barrier(0);  // requests first, then the MFMAs they fly under
            if (kc < 8) WF_STAMP(5 + 3 * kc);
            multiply(cur, early ? -1 : kc, kc == 0, kc);
            __builtin_amdgcn_sched_barrier(0);
            if (kc < 8) WF_STAMP(6 + 3 * kc);
            if (kc + 2 < nchunks) dma_wait_n<8>(); else dma_wait();  // uniform; 8 = the loads of one load_patch
            __syncthreads();
        }

        // ---- epilogue, with the next unit's chunk 0 started underneath it ---------------------------------
        WF_STAMP(28);
        const bool e_whole = whole, e_tile_ok = tile_ok;
        const int e_m0 = m0, e_tb = tb, e_half = half, oh = 2 * th, ow = 2 * tw, e_slot = slot;
        const unsigned e_n = n;
        const bool has_next = it + 1 < nitems;
        if (has_next) start_unit(it + 1);  // both stages are free: the K loop ended with a barrier
        const bool two_cols = ow + 1 < a.W, two_rows = oh + 1 < a.H;
        // byte offsets of this lane's 2 x 2 outputs in channel 0 of its image; out of range = not stored
        const unsigned o00 = e_tile_ok ? (e_n * (unsigned)a.M * (unsigned)HW + (unsigned)(oh * a.W + ow)) * 4u : kOOB;
        const unsigned o10 = (e_tile_ok && two_rows) ? o00 + (unsigned)a.W * 4u : kOOB;
        const unsigned o01 = two_cols ? o00 + 4u : kOOB, o11 = two_cols ? o10 + 4u : kOOB;  // used when W is odd
        const float w00 = e_tile_ok ? 1.f : 0.f, w10 = (e_tile_ok && two_rows) ? 1.f : 0.f, w01 = two_cols ? 1.f : 0.f;
        const unsigned st_voff = lane == 63 ? 0u : kOOB, st_voff2 = (lane == 63 && e_whole) ? 8u : kOOB;
        float* const S = lds + WF_STAGE;   // [4 rows of positions][2][32 channels][64 tiles]: stage 1 only
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {   // the 64 channels in two halves of 32
            if (ph == 1) __syncthreads();  // the first half's readers are done with S
            // (1) column half of A^T m A on the accumulators: S[b] = sum_j m[j] * A[j][b], A^T = [1 1 1 0; 0 1 -1 -1]
            if (fh == ph) {                // uniform: the four waves that hold this half
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    if (tt == 1 && !e_whole) break;  // uniform
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {  // register pairs: packed fp32 adds
                        const buf_f32x2 m0v = {acc[0][tt][r], acc[0][tt][r + 1]}, m1v = {acc[1][tt][r], acc[1][tt][r + 1]};
                        const buf_f32x2 m2v = {acc[2][tt][r], acc[2][tt][r + 1]}, m3v = {acc[3][tt][r], acc[3][tt][r + 1]};
                        const buf_f32x2 s0 = m0v + m1v + m2v, s1 = m1v - m2v - m3v;
                        float* p = S + ((wrow * 2) * 32 + mfma_row(r, lane)) * WF_BT + tt * 32 + l31;  // rows r, r + 1 are adjacent
                        p[0] = s0[0];
                        p[WF_BT] = s0[1];
                        p[32 * WF_BT] = s1[0];
                        p[33 * WF_BT] = s1[1];
                    }
                }
            }
            if (ph == 0) WF_STAMP(29);
            __syncthreads();
            if (ph == 0) WF_STAMP(30);
            // (2) one (channel, tile) per lane: the row half, bias / activation, stores, statistics. Branch-free (stores are
            //     masked by address) so that the four channels of a wave overlap their LDS reads, adds and stores.
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int fl = q * 8 + wid;  // wave-uniform channel of this half
                const int f = e_m0 + ph * 32 + fl;
                const bool f_ok = f < a.M;   // uniform
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
                    if (bv == 1.0f) bv = 0.f;  // bcnn_add_scalar of the AVX build adds nothing for exactly 1.0f (quirk 2)
                    const float sl = (EPI == 2 && a.act == BCNN_HIP_ACT_PRELU && a.slopes && f_ok) ? a.slopes[f] : 0.f;
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            float vv = o[r][c];
                            if (bv != 0.0f) vv += bv;
                            if (EPI == 1) vv = vv * (float)(vv > 0);  // RELU as the reference's multiply (-0.0f, NaN propagate)
                            else if (a.act != BCNN_HIP_ACT_NONE) vv = act_fwd_cheap(vv, a.act, sl);
                            o[r][c] = vv;
                        }
                }
                const int soff = f_ok ? f * HW * 4 : 0;  // wave-uniform; soffset is not range-checked,
                const unsigned kill = f_ok ? 0u : kOOB;  // ... the voffset is: channels past M are dropped through it
                if (EPI == 0 && e_slot >= 0) {  // uniform: a piece of a K-split unit -- raw partial outputs to its scratch slot
                    const unsigned so = (unsigned)(((e_slot * 64 + ph * 32 + fl) * 64 + lane) * 16);
                    buffer_store_f32x2(buf_f32x2{o[0][0], o[0][1]}, rs_scr, (int)so, 0, 0);
                    buffer_store_f32x2(buf_f32x2{o[1][0], o[1][1]}, rs_scr, (int)(so + 8u), 0, 0);
                    continue;
                }
                if (!odd_w) {  // uniform; rows start 8-byte aligned and ow is even
                    buffer_store_f32x2(buf_f32x2{o[0][0], o[0][1]}, rs_dst, (int)(o00 | kill), soff, 0);
                    buffer_store_f32x2(buf_f32x2{o[1][0], o[1][1]}, rs_dst, (int)(o10 | kill), soff, 0);
                } else {
                    buffer_store_f32(o[0][0], rs_dst, (int)(o00 | kill), soff, 0);
                    buffer_store_f32(o[0][1], rs_dst, (int)(o01 | kill), soff, 0);
                    buffer_store_f32(o[1][0], rs_dst, (int)(o10 | kill), soff, 0);
                    buffer_store_f32(o[1][1], rs_dst, (int)(o11 | kill), soff, 0);
                }
                if (STATS) {  // this wave holds channel f for the workgroup's 64 tiles
                    // masked by multiplication: lanes without a tile / a second row / a second column add 0.0
                    const float sv0 = (o[0][0] + o[0][1] * w01) * w00, sv1 = (o[1][0] + o[1][1] * w01) * w10;
                    const float sq0 = (o[0][0] * o[0][0] + o[0][1] * o[0][1] * w01) * w00;
                    const float sq1 = (o[1][0] * o[1][0] + o[1][1] * o[1][1] * w01) * w10;
                    const float sv = wave_sum_dpp(sv0 + sv1), sq = wave_sum_dpp(sq0 + sq1);
                    // slot per half block; a whole block owns both and zeroes the second (lane 63 holds the sums)
                    const int st_soff = f_ok ? (f * (2 * a.tblocks) + 2 * e_tb + e_half) * 8 : 0;
                    buffer_store_f32x2(buf_f32x2{sv, sq}, rs_stats, (int)(st_voff | kill), st_soff, 0);
                    buffer_store_f32x2(buf_f32x2{0.f, 0.f}, rs_stats, (int)(st_voff2 | kill), st_soff, 0);
                }
            }
            // the next unit's first patches have had the first half of the epilogue to arrive
            if (ph == 0 && has_next) finish_start();
        }
        WF_STAMP(31);
    }
}

// K-split tail of wino_fused_kernel<0, .>: the pieces of tail unit u (unit index nunits + u) sit in the scratch slots of the
// workgroups whose chunk ranges met it -- workgroup i's range starts at chunk i * tail_q of the flattened tail; a range that
// started in the previous unit left its SECOND piece here (slot 2 i + 1), every other one its first (slot 2 i). They are
// added in channel order (ascending i), then stored and counted into the batch-norm statistics exactly like the epilogue of
// the main kernel does for a whole unit. One wave per (unit, channel): lane = tile.
template <bool STATS>
__global__ __launch_bounds__(256) void wino_tail_fixup_kernel(const WinoFusedArgs a) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int u = blockIdx.x >> 4, fl = (blockIdx.x & 15) * 4 + wid;
    const int NC = a.Jpad / WF_KC;
    const int blk = a.nunits + u;
    const int mb = blk % a.mblocks, tb = blk / a.mblocks;
    const int f = mb * WF_BF + fl;
    const bool f_ok = f < a.M;
    const unsigned t = (unsigned)tb * WF_BT + (unsigned)lane;
    const bool tile_ok = t < a.T;
    const unsigned per_img = (unsigned)(a.TH * a.TW);
    const unsigned n = tile_ok ? t / per_img : 0u;
    const unsigned rr = tile_ok ? t - n * per_img : 0u;
    const int th = (int)(rr / (unsigned)a.TW), tw = (int)(rr - (unsigned)th * (unsigned)a.TW);
    const int c0 = u * NC, c1 = c0 + NC;
    const int i0 = c0 / a.tail_q, i1 = (c1 - 1) / a.tail_q;
    float o00 = 0.f, o01 = 0.f, o10 = 0.f, o11 = 0.f;
    for (int i = i0; i <= i1; ++i) {
        const int slot = 2 * i + (i * a.tail_q < c0 ? 1 : 0);
        const float4 v = *reinterpret_cast<const float4*>(a.tail_scr + ((size_t)(slot * 64 + fl) * 64 + lane) * 4);
        o00 += v.x; o01 += v.y; o10 += v.z; o11 += v.w;
    }
    const int HW = a.H * a.W, oh = 2 * th, ow = 2 * tw;
    const bool two_cols = ow + 1 < a.W, two_rows = oh + 1 < a.H;
    if (tile_ok && f_ok) {
        float* d = a.dst + ((size_t)n * a.M + f) * HW + (size_t)oh * a.W + ow;
        d[0] = o00;
        if (two_cols) d[1] = o01;
        if (two_rows) {
            d[a.W] = o10;
            if (two_cols) d[a.W + 1] = o11;
        }
    }
    if (STATS) {  // the epilogue's masked sums, in its order
        const float w00 = tile_ok ? 1.f : 0.f, w10 = (tile_ok && two_rows) ? 1.f : 0.f, w01 = two_cols ? 1.f : 0.f;
        const float sv0 = (o00 + o01 * w01) * w00, sv1 = (o10 + o11 * w01) * w10;
        const float sq0 = (o00 * o00 + o01 * o01 * w01) * w00, sq1 = (o10 * o10 + o11 * o11 * w01) * w10;
        const float sv = wave_sum_dpp(sv0 + sv1), sq = wave_sum_dpp(sq0 + sq1);
        if (lane == 63 && f_ok) {  // a whole block owns both half-block slots and zeroes the second
            float* st = a.stats + ((size_t)f * (2 * a.tblocks) + 2 * tb) * 2;
            st[0] = sv; st[1] = sq; st[2] = 0.f; st[3] = 0.f;
        }
    }
}

// U[xi][j][m] = (G g G^T)[xi] packed [16][Jpad][Mpad] with zero padding.
//   forward: m = f, j = c; dX: m = c, j = f and the filter rotated by 180 degrees
__device__ __forceinline__ void wino_pack_one(const float* __restrict__ w, float* __restrict__ u, int F, int C, int dx_mode,
                                              int Jpad, int Mpad, int idx) {
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
    const size_t plane = (size_t)Jpad * Mpad;
    float* dst = u + idx;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        dst[(size_t)(4 * r + 0) * plane] = t[r][0];
        dst[(size_t)(4 * r + 1) * plane] = 0.5f * (t[r][0] + t[r][1] + t[r][2]);
        dst[(size_t)(4 * r + 2) * plane] = 0.5f * (t[r][0] - t[r][1] + t[r][2]);
        dst[(size_t)(4 * r + 3) * plane] = t[r][2];
    }
}

__global__ __launch_bounds__(256) void wino_pack_weights_kernel(const float* __restrict__ w, float* __restrict__ u, int F,
                                                                int C, int dx_mode, int Jpad, int Mpad) {
    wino_pack_one(w, u, F, C, dx_mode, Jpad, Mpad, blockIdx.x * 256 + threadIdx.x);
}

// the transformed weights of many layers in one launch (bcnn_hip_conv_prepack): blockIdx.y = job
__global__ __launch_bounds__(256) void wino_pack_weights_multi_kernel(const WinoPackJob* __restrict__ jobs) {
    const WinoPackJob j = jobs[blockIdx.y];
    if ((int)blockIdx.x >= j.blocks) return;
    if (j.npos == 36) wino43_pack_one(j.w, j.u, j.F, j.C, j.dx_mode, j.Jpad, j.Mpad, blockIdx.x * 256 + threadIdx.x, j.layout);
    else wino_pack_one(j.w, j.u, j.F, j.C, j.dx_mode, j.Jpad, j.Mpad, blockIdx.x * 256 + threadIdx.x);
}

// ---- host side ------------------------------------------------------------------------------------------
struct WfScratch {
    float* p = nullptr;
    size_t cap = 0;
    int dev = -1;
};
static thread_local WfScratch g_wf_scratch;
static float* wf_scratch(size_t floats) {
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    WfScratch& sc = g_wf_scratch;
    if (sc.p == nullptr || sc.cap < floats || sc.dev != dev) {
        if (sc.p && sc.dev == dev) {
            HIP_CHECK(hipStreamSynchronize(current_stream()));
            HIP_CHECK(hipFree(sc.p));
        }
        const size_t cap = floats < (1u << 20) ? (1u << 20) : floats;
        HIP_CHECK(hipMalloc((void**)&sc.p, cap * sizeof(float)));
        sc.cap = cap;
        sc.dev = dev;
    }
    return sc.p;
}

// the piece outputs of a K-split tail (33.5 MB: two slots of 64 x 64 x 4 floats per CU); separate from the U scratch, which
// the running kernel reads
static thread_local WfScratch g_wf_tail_scratch;
static float* wf_tail_scratch(size_t floats) {
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    WfScratch& sc = g_wf_tail_scratch;
    if (sc.p == nullptr || sc.cap < floats || sc.dev != dev) {
        if (sc.p && sc.dev == dev) {
            HIP_CHECK(hipStreamSynchronize(current_stream()));
            HIP_CHECK(hipFree(sc.p));
        }
        HIP_CHECK(hipMalloc((void**)&sc.p, floats * sizeof(float)));
        sc.cap = floats;
        sc.dev = dev;
    }
    return sc.p;
}

#ifdef BCNN_HIP_EXPERIMENT
}  // namespace bcnn_hip
#include "wino_bf16_exp.h"  // the split-bf16 experiment (kernels + launch): not part of the product library
namespace bcnn_hip {
#else
constexpr int WB_KC = WF_KC;                    // (only the experiment's kernels have a chunk size of their own)
static int wino_bf16_parts(int) { return 0; }  // the product library carries no bf16 path
#endif

static int g_wf_force = -1;  // experiment build: BCNN_HIP_WINOGRAD_FUSED=0/1 overrides the rule
static bool wino_fused_wanted(const ConvShape& s, int J, int M) {
    if (s.ksz != 3 || s.stride != 1 || s.pad != 1 || s.groups != 1) return false;
    if (J < 16 || (J % WF_KC) != 0 || M < 32) return false;
    if ((size_t)s.N * J * s.HW * 4 >= 0x7ffffff0ull || (size_t)s.N * M * s.HW * 4 >= 0x7ffffff0ull) return false;
    if (g_wf_force < 0) {
        const char* e = BCNN_EXP_ENV("BCNN_HIP_WINOGRAD_FUSED");
        g_wf_force = e ? (e[0] == '0' ? 0 : 1) : 2;
    }
    if (g_wf_force != 2) return g_wf_force == 1;
    // Measured on the ResNet-18 shapes at N = 128 (tools/exp/wino_sweep.sh): 1.2-1.6x faster than
    // the direct LDS-DMA kernels from 64 channels up, and as fast as or faster than the three-kernel form
    // (conv_winograd.hip) on the deep layers -- so it takes every eligible layer with enough tiles to fill the chip.
    const long long T = (long long)s.N * ((s.H + 1) / 2) * ((s.W + 1) / 2);
    return J >= 64 && M >= 64 && T * ((M + WF_BF - 1) / WF_BF) >= (long long)WF_BT * kCUs / 2;
}

static void wino_fused_run(const float* src, const float* w, float* dst, const ConvShape& s, int dx_mode,
                           const float* bias, const float* slopes, int act, int add_bias, ConvStats* stats) {
    WinoFusedArgs a;
    a.src = src; a.dst = dst; a.bias = bias; a.slopes = slopes;
    a.N = s.N; a.J = dx_mode ? s.F : s.C; a.M = dx_mode ? s.C : s.F; a.H = s.H; a.W = s.W;
    a.TH = (s.H + 1) / 2; a.TW = (s.W + 1) / 2;
    a.T = (unsigned)((long long)s.N * a.TH * a.TW);
    const int parts = wino_bf16_parts(a.J);
    a.Jpad = parts ? (a.J + WB_KC - 1) / WB_KC * WB_KC : (a.J + WF_KC - 1) / WF_KC * WF_KC;
    a.Mpad = (a.M + WF_BF - 1) / WF_BF * WF_BF;
    a.mblocks = a.Mpad / WF_BF;
    a.tblocks = (int)((a.T + WF_BT - 1) / WF_BT);
    a.act = act; a.add_bias = add_bias;
    a.src_bytes = (unsigned)((size_t)s.N * a.J * s.HW * 4);
    a.dst_bytes = (unsigned)((size_t)s.N * a.M * s.HW * 4);
    if (!parts) {
        const size_t u_floats = (size_t)16 * a.Jpad * a.Mpad;
        a.upk_bytes = (unsigned)(u_floats * 4);
        float* U = prepack_take(w, PREPACK_WINO, dx_mode, u_floats);  // transformed ahead by bcnn_hip_conv_prepack?
        if (!U) {
            U = wf_scratch(u_floats);
            wino_pack_weights_kernel<<<ceil_div((long long)a.Jpad * a.Mpad, 256), 256, 0, current_stream()>>>(w, U, s.F, s.C, dx_mode,
                                                                                                            a.Jpad, a.Mpad);
            KERNEL_CHECK();
        }
        a.upk = U;
    }
    a.stats = (stats && stats->partials) ? stats->partials : nullptr;
    const int nblocks = a.tblocks * a.mblocks;
    const unsigned grid = (unsigned)(nblocks < kCUs ? nblocks : kCUs);  // persistent: one 128 KB workgroup per CU
    // the last round's R blocks as 2R half blocks when that fills no more than one round
    const int rem = nblocks % (int)grid;
    a.nfull = (rem > 0 && 2 * rem <= (int)grid && nblocks > (int)grid) ? nblocks - rem : nblocks;
    a.nunits = a.nfull + 2 * (nblocks - a.nfull);
    const bool plain = !a.add_bias && a.act == BCNN_HIP_ACT_NONE;
    // ... or, for the raw-output kernels, that round's chunks dealt out evenly over all CUs (K-split tail): a CU then does
    // ceil(R * chunks / CUs) chunks in one or two pieces instead of a whole (or half) unit. Priced in chunk times with ~1.5
    // per epilogue; the fix-up pass over the tail's outputs is a few microseconds.
    a.tail_scr = nullptr; a.tail_q = 0; a.tail_units = 0; a.tail_scr_bytes = 0;
    static const int ksplit_on = BCNN_EXP_ENV("BCNN_HIP_NO_WINO_KSPLIT") ? 0 : 1;  // A/B switch of the experiment build
    if (ksplit_on && plain && !parts && rem > 0 && nblocks > (int)grid) {
        const int NC = a.Jpad / WF_KC;
        const int q = (int)ceil_div((long long)rem * NC, (long long)grid);
        // (+2: the fix-up launch, 5-8 us -- on the 64-channel 56 x 56 layers the two ways then cost the same: not split)
        const double ep = 1.5, now = (a.nfull < nblocks ? 0.5 * NC : (double)NC) + ep, split = q + 2 * ep + 2.0;
        if (q >= 1 && q <= NC && split < now) {
            a.nfull = nblocks;         // every block is a whole unit ...
            a.nunits = nblocks - rem;  // ... and the last `rem` of them are the tail
            a.tail_units = rem;
            a.tail_q = q;
            const size_t scr_floats = (size_t)2 * grid * 64 * 64 * 4;
            a.tail_scr = wf_tail_scratch(scr_floats);
            a.tail_scr_bytes = (unsigned)(scr_floats * sizeof(float));
        }
    }
    a.stats_bytes = a.stats ? (unsigned)((size_t)a.M * 2 * a.tblocks * 2 * sizeof(float)) : 0u;
    if (a.stats && (size_t)a.stats_bytes > stats->capacity * sizeof(float)) {
        fprintf(stderr, "[bcnn_hip] fused Winograd: statistics need %u bytes, the caller's buffer holds %zu\n",
                a.stats_bytes, stats->capacity * sizeof(float));
        abort();
    }
    if (a.stats && !plain) {
        fprintf(stderr, "[bcnn_hip] fused Winograd: output statistics are taken on the raw convolution output only\n");
        abort();
    }
    trace_kernel(dx_mode ? "wino_fused_kernel:dx" : "wino_fused_kernel:fwd");
#ifdef BCNN_HIP_EXPERIMENT
    if (parts == 2) { wino_bf16_launch<2>(a, w, s, dx_mode, grid, plain); }
    else if (parts == 3) { wino_bf16_launch<3>(a, w, s, dx_mode, grid, plain); }
    else
#endif
    if (a.stats) wino_fused_kernel<0, true><<<grid, 512, 0, current_stream()>>>(a);
    else if (plain) wino_fused_kernel<0, false><<<grid, 512, 0, current_stream()>>>(a);
    else if (a.act == BCNN_HIP_ACT_RELU) wino_fused_kernel<1, false><<<grid, 512, 0, current_stream()>>>(a);
    else wino_fused_kernel<2, false><<<grid, 512, 0, current_stream()>>>(a);
    KERNEL_CHECK();
    if (a.tail_units > 0) {
        trace_kernel("wino_tail_fixup");
        if (a.stats) wino_tail_fixup_kernel<true><<<(unsigned)(a.tail_units * 16), 256, 0, current_stream()>>>(a);
        else wino_tail_fixup_kernel<false><<<(unsigned)(a.tail_units * 16), 256, 0, current_stream()>>>(a);
        KERNEL_CHECK();
    }
    if (stats) stats->splits = a.stats ? 2 * a.tblocks : 0;
}

// bcnn_hip_conv_prepack: the transformed weights this layer's forward (dx_mode 0) / data-gradient (1) kernel will ask
// prepack_take for; false when the layer does not run on wino_fused_kernel
bool wino_fused_pack_plan(const ConvShape& s, int dx_mode, WinoPackJob* job, size_t* floats) {
    const int J = dx_mode ? s.F : s.C, M = dx_mode ? s.C : s.F;
    // the F(4x4,3x3) kernel is asked first by both passes (its forward only in the raw form: a layer without a batch-norm
    // behind it then packs for itself, like every layer whose planned kernel does not run)
    if (wino43_pack_plan(s, dx_mode, job, floats)) return true;
    if (!wino_fused_wanted(s, J, M) || wino_bf16_parts(J) != 0) return false;
    job->w = nullptr; job->u = nullptr;
    job->F = s.F; job->C = s.C; job->dx_mode = dx_mode;
    job->Jpad = (J + WF_KC - 1) / WF_KC * WF_KC;
    job->Mpad = (M + WF_BF - 1) / WF_BF * WF_BF;
    job->blocks = (int)ceil_div((long long)job->Jpad * job->Mpad, 256);
    job->npos = 16;
    job->layout = 0;
    *floats = (size_t)16 * job->Jpad * job->Mpad;
    return true;
}

void wino_fused_pack_launch(const WinoPackJob* jobs_dev, int n, int max_blocks) {
    wino_pack_weights_multi_kernel<<<dim3((unsigned)max_blocks, (unsigned)n), 256, 0, current_stream()>>>(jobs_dev);
    KERNEL_CHECK();
}

static double wf_flops(const ConvShape& s) {
    const double T = (double)s.N * ((s.H + 1) / 2) * ((s.W + 1) / 2);
    return 2.0 * 16.0 * T * s.C * s.F;
}
// the same without the tiles' overhang on odd-sized planes (7 x 7: 16 tiles cover 8 x 8)
static double wf_useful_flops(const ConvShape& s) { return 2.0 * 16.0 * ((double)s.N * s.H * s.W / 4.0) * s.C * s.F; }
static double wf_bytes(const ConvShape& s) {
    return 4.0 * ((double)s.N * s.C * s.HW + (double)s.F * s.K + (double)s.N * s.F * s.OHOW);
}

bool conv_forward_winograd_fused(const float* x, const float* w, const float* bias, const float* slopes, float* y,
                                 const ConvShape& s, int act, int raw, ConvStats* stats) {
    if (!wino_fused_wanted(s, s.C, s.F)) return false;
    KTimer kt(K_CONV_FWD_WINO, wf_flops(s), wf_bytes(s), wf_useful_flops(s));
    if (stats && !raw) stats->splits = 0;
    wino_fused_run(x, w, y, s, 0, bias, slopes, raw ? BCNN_HIP_ACT_NONE : act, raw ? 0 : (bias != nullptr), raw ? stats : nullptr);
    return true;
}

bool conv_backward_data_winograd_fused(const float* w, const float* dy, float* dx, const ConvShape& s) {
    if (!wino_fused_wanted(s, s.F, s.C)) return false;
    KTimer kt(K_CONV_DX_WINO, wf_flops(s), wf_bytes(s), wf_useful_flops(s));
    wino_fused_run(dy, w, dx, s, 1, nullptr, nullptr, BCNN_HIP_ACT_NONE, 0, nullptr);
    return true;
}


// =============================================================================================================
// Weight gradient in the transformed domain, one kernel:  dU[xi][f][c] = sum_t dM[xi][f][t] * V[xi][c][t]
//   dM = A dy_tile A^T (adjoint of the output transform), V = B^T x_patch B; then dw += G^T dU G (finalize kernel).
// A GEMM per position with the reduction over ALL tiles of the batch: a workgroup owns a 64 x 64 (f, c) block of
// all 16 positions (8 waves x 2 positions x 2x2 accumulator tiles = 128 registers) and a contiguous range of tiles;
// per chunk of 8 tiles every thread transforms one (f, tile) dy block and one (c, tile) x patch straight from
// global memory into LDS ([xi][channel][tile], rows padded to 9 floats: conflict-free fragment reads), requested
// before the chunk's 32 MFMAs per wave and written after them. Partial blocks go to the workspace and are added to
// dw in a fixed order (deterministic; keeps the `+=` onto the momentum carry, bcnn_conv_layer.c:547-553).
// =============================================================================================================
constexpr int WD_KT = 8;                 // tiles per chunk
// LDS rows are the 8 tiles of a chunk, unpadded, with the tile index XOR-swizzled by the channel: element (xi, ch, t) sits
// at (xi * 64 + ch) * 8 + (t ^ swz(ch)), swz(ch) = 2 * ((ch >> 3) & 3) + ((ch >> 2) & 1). ds_read_b32 / ds_write_b32 are
// serviced per 32-lane half with bank = (address / 4) mod 32 (MI355X_MICROARCH.md, LDS): the transform's writes (a half-wave =
// 8 tiles x 4 consecutive channels with one value of (ch >> 2) & 1 and of (ch >> 3) & 3: ch * 8 + t' covers 32 consecutive
// floats) and the MFMA fragment reads (a half-wave = channels l31 = a + 8 b, one tile 2 ks + lhi: bank 8 (a & 3) + (t ^ swz),
// and the eight combinations of (a >> 2, b) give eight different swz) both touch 32 different banks. Round 4's swizzle
// (2 * ((ch >> 3) & 3) alone, reasoned on 64 banks) left channels a and a + 4 of a fragment read on one bank: the 2-way
// conflict rocprofv3 kept counting as 33 % of the kernel's LDS cycles (profiles/r04_sq_step_resnet18.txt). The padded 9-float
// rows of round 3 took 147 KB where this takes 128.
constexpr int WD_ROW = WD_KT;            // LDS row: the chunk's 8 tiles
constexpr int WD_OP = 16 * 64 * WD_ROW;  // floats per operand and stage


struct WinoDwArgs {
    const float* x;    // [N][C][H][W]
    const float* dy;   // [N][F][H][W]
    float* partials;   // [splits][fblocks * cblocks][16][64][64]
    int N, C, F, H, W, TH, TW;
    unsigned T, tiles_per_split;
    int fblocks, cblocks, splits;
    unsigned x_bytes, dy_bytes;
};

template <bool ODDW>
__global__ __launch_bounds__(512, 2) void wino_dw_fused_kernel(const WinoDwArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 2 * WD_OP];  // two stages of (dM, V): 128 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int nob = a.fblocks * a.cblocks;
    const int ob = blockIdx.x % nob, sp = blockIdx.x / nob;
    const int f0 = (ob % a.fblocks) * 64, c0 = (ob / a.fblocks) * 64;
    const unsigned tbeg = (unsigned)sp * a.tiles_per_split;
    unsigned tend = tbeg + a.tiles_per_split;
    if (tend > a.T) tend = a.T;
    const int nchunks = tbeg < tend ? (int)((tend - tbeg + WD_KT - 1) / WD_KT) : 0;

    // this thread's transform item in every chunk: channel ch (of both the f block and the c block), tile tl
    const int tl = lane & 7, ch = wid * 8 + (lane >> 3);
    const bool f_ok = f0 + ch < a.F, c_ok = c0 + ch < a.C;
    const bool is_l = tl == 0, is_r = tl == 7;  // the lanes whose left / right neighbour tile sits in no lane of the wave
    // tile coordinates of (tbeg + tl), advanced by 8 tiles per chunk; xo / yo = byte offsets of x[n][c0+ch][2th][2tw] and
    // dy[n][f0+ch][2th][2tw], advanced with them (no multiplications in the loop)
    unsigned t = tbeg + (unsigned)tl;
    const unsigned per_img = (unsigned)(a.TH * a.TW);
    int th, tw;
    unsigned xo, yo;
    {
        const int n = (int)(t / per_img);
        const unsigned rr = t - (unsigned)n * per_img;
        th = (int)(rr / (unsigned)a.TW);
        tw = (int)(rr - (unsigned)th * (unsigned)a.TW);
        xo = (unsigned)(((n * a.C + c0 + ch) * a.H + 2 * th) * a.W + 2 * tw) * 4u;
        yo = (unsigned)(((n * a.F + f0 + ch) * a.H + 2 * th) * a.W + 2 * tw) * 4u;
    }
    const unsigned W4 = (unsigned)a.W * 4u;
    const unsigned step_row = (unsigned)(2 * a.W - 2 * a.TW) * 4u;                      // one row carry
    const unsigned step_img_x = (unsigned)(a.C * a.H * a.W - 2 * a.W * a.TH) * 4u;      // one image carry
    const unsigned step_img_y = (unsigned)(a.F * a.H * a.W - 2 * a.W * a.TH) * 4u;
    const rsrc_i4 rs_x = make_rsrc(a.x, a.x_bytes), rs_y = make_rsrc(a.dy, a.dy_bytes);

    constexpr bool odd_w = ODDW;
    // x patch rows as requested: the pair of columns (2tw, 2tw+1), the left neighbour column (lanes tl == 0 only) and the
    // right one (lanes tl == 7 only); dy block rows as pairs
    buf_f32x2 xm[4], gy[2];
    float xl[4], xr[4];
    bool sel_l = false, sel_r = false, pair_y_ok = true;
    auto load_items = [&]() {
        // Rows 2th-1 .. 2th+2, columns 2tw-1 .. 2tw+2 of channel c0 + ch, zero outside the image: every load is a raw
        // buffer load whose voffset is out of range (-> 0.0, no memory access) when the row, the channel or the tile
        // does not exist; a pair never leaves its row (for odd W its second element can, and is then masked). The
        // neighbour columns of tiles 1..6 of the 8-tile group are the neighbouring lanes' pairs (DPP row shifts in
        // write_items); only the group's first / last lane fetch theirs.
        const bool live = t < tend;
        const bool x_ok = live && c_ok, y_ok = live && f_ok;
        const bool r0 = th > 0, r2 = 2 * th + 1 < a.H, r3 = 2 * th + 2 < a.H;
        const bool has_l = tw > 0, has_r = tw + 1 < a.TW;
        sel_l = !is_l && has_l;
        sel_r = !is_r && has_r;
        if (odd_w) pair_y_ok = 2 * tw + 1 < a.W;
        const bool el = x_ok && is_l && has_l, er = x_ok && is_r && has_r;
        const unsigned up = xo - W4;
        // the scalar offsets ARE wave-uniform, but when the compiler keeps W4 in a vector register (it runs out of scalar
        // ones here) it wraps every such load in a waterfall loop over the distinct values: four loops of ~10 instructions
        // per chunk. readfirstlane tells it.
        const int s1 = __builtin_amdgcn_readfirstlane((int)W4), s2 = __builtin_amdgcn_readfirstlane((int)(2u * W4));
        xm[0] = buffer_load_f32x2(rs_x, (int)((x_ok && r0) ? up : kOOB), 0, 0);
        xm[1] = buffer_load_f32x2(rs_x, (int)(x_ok ? xo : kOOB), 0, 0);
        xm[2] = buffer_load_f32x2(rs_x, (int)((x_ok && r2) ? xo : kOOB), s1, 0);
        xm[3] = buffer_load_f32x2(rs_x, (int)((x_ok && r3) ? xo : kOOB), s2, 0);
        xl[0] = buffer_load_f32(rs_x, (int)((el && r0) ? up - 4u : kOOB), 0, 0);
        xl[1] = buffer_load_f32(rs_x, (int)(el ? xo - 4u : kOOB), 0, 0);
        xl[2] = buffer_load_f32(rs_x, (int)((el && r2) ? xo - 4u : kOOB), s1, 0);
        xl[3] = buffer_load_f32(rs_x, (int)((el && r3) ? xo - 4u : kOOB), s2, 0);
        xr[0] = buffer_load_f32(rs_x, (int)((er && r0) ? up + 8u : kOOB), 0, 0);
        xr[1] = buffer_load_f32(rs_x, (int)(er ? xo + 8u : kOOB), 0, 0);
        xr[2] = buffer_load_f32(rs_x, (int)((er && r2) ? xo + 8u : kOOB), s1, 0);
        xr[3] = buffer_load_f32(rs_x, (int)((er && r3) ? xo + 8u : kOOB), s2, 0);
        // dy block: rows 2th, 2th+1, columns 2tw, 2tw+1 of channel f0 + ch
        gy[0] = buffer_load_f32x2(rs_y, (int)(y_ok ? yo : kOOB), 0, 0);
        gy[1] = buffer_load_f32x2(rs_y, (int)((y_ok && r2) ? yo : kOOB), s1, 0);
    };
    auto advance = [&]() {  // 8 tiles on: at most two row carries (TW >= 4) and one image carry (TH >= 2)
        t += WD_KT;
        tw += WD_KT;
        const bool c1 = tw >= a.TW, c2 = tw >= 2 * a.TW;
        const int rows = (c1 ? 1 : 0) + (c2 ? 1 : 0);
        tw -= rows * a.TW;
        th += rows;
        const bool ci = th >= a.TH;
        if (ci) th -= a.TH;
        const unsigned d = 64u + (c1 ? step_row : 0u) + (c2 ? step_row : 0u);
        xo += d + (ci ? step_img_x : 0u);
        yo += d + (ci ? step_img_y : 0u);
    };
    float d[4][4], g[2][2];
    auto assemble = [&]() {  // the requested registers -> the 4 x 4 patch and the 2 x 2 dy block
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float m0 = xm[i][0], m1 = xm[i][1];
            if (odd_w) m1 = pair_y_ok ? m1 : 0.f;
            // (bound_ctrl: a row's first / last lane reads 0 -- those lanes take xl / xr below anyway -- and the move needs no
            // initialised destination, i.e. no v_mov in front of it)
            const float fromleft = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, m1), 0x111 /* row_shr:1 */, 0xf, 0xf, true));
            const float fromright = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, m0), 0x101 /* row_shl:1 */, 0xf, 0xf, true));
            d[i][0] = sel_l ? fromleft : xl[i];
            d[i][1] = m0;
            d[i][2] = m1;
            d[i][3] = sel_r ? fromright : xr[i];
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            g[r][0] = gy[r][0];
            g[r][1] = odd_w ? (pair_y_ok ? gy[r][1] : 0.f) : gy[r][1];
        }
    };
    auto write_items = [&](int stage) {
        assemble();
        float* pm = lds + stage * 2 * WD_OP + ch * WD_ROW + (tl ^ (2 * (wid & 3) + lhi));  // (ch >> 3) & 3 == wid & 3, (ch >> 2) & 1 == lhi
        float* pv = pm + WD_OP;
        // dM = A g A^T, A = [1 0; 1 1; 1 -1; 0 -1]
        float q[4][2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            q[0][c] = g[0][c];
            q[1][c] = g[0][c] + g[1][c];
            q[2][c] = g[0][c] - g[1][c];
            q[3][c] = -g[1][c];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            pm[(4 * i + 0) * 64 * WD_ROW] = q[i][0];
            pm[(4 * i + 1) * 64 * WD_ROW] = q[i][0] + q[i][1];
            pm[(4 * i + 2) * 64 * WD_ROW] = q[i][0] - q[i][1];
            pm[(4 * i + 3) * 64 * WD_ROW] = -q[i][1];
        }
        // V = B^T d B
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
            pv[(4 * i + 0) * 64 * WD_ROW] = tt[i][0] - tt[i][2];
            pv[(4 * i + 1) * 64 * WD_ROW] = tt[i][1] + tt[i][2];
            pv[(4 * i + 2) * 64 * WD_ROW] = tt[i][2] - tt[i][1];
            pv[(4 * i + 3) * 64 * WD_ROW] = tt[i][1] - tt[i][3];
        }
    };

    f32x16 acc[2][2][2];  // [position of the pair][f tile][c tile]
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[e][i][j][r] = 0.f;

    // The vector ALU (transform) and the fp32 MFMAs do not overlap on a SIMD, but each SIMD hosts one wave of each half of
    // the workgroup, and the halves transform at different times -- waves 0-3 at the head of a chunk, waves 4-7 between
    // their second and third k-step -- so that while one wave waits for its loads, its LDS stores or the barrier, the other
    // has MFMAs to issue (the arrangement of wino_fused_kernel). The items a wave transforms in chunk kc (for chunk kc + 1)
    // were requested during chunk kc - 1, right after the previous transform freed the registers: a whole chunk of lead.
    const bool early = wid < 4;
    const unsigned rd0 = (unsigned)(((2 * wid) * 64 + l31) * WD_ROW + (lhi ^ ((l31 >> 2) & 1)) + 2 * ((l31 >> 3) & 3));  // fragment reads, see WD_ROW
    if (nchunks > 0) {
        load_items();
        write_items(0);
        if (nchunks > 1) {
            advance();
            load_items();
        }
    }
    __syncthreads();
    for (int kc = 0; kc < nchunks; ++kc) {
        const int cur = kc & 1, nxt = cur ^ 1;
        const bool more = kc + 1 < nchunks, more2 = kc + 2 < nchunks;
        auto produce = [&]() {  // V / dM of chunk kc + 1 from the registers, then the requests of chunk kc + 2 into them
            if (more) write_items(nxt);
            if (more2) {
                advance();
                load_items();
            }
        };
        if (early) produce();
        __builtin_amdgcn_sched_barrier(0);  // requests first, then the MFMAs they fly under
        const float* stage_base = lds + cur * 2 * WD_OP;
#pragma unroll
        for (int ks = 0; ks < WD_KT / 2; ++ks) {
            float af[2][2], bf[2][2];
            // tile 2 ks + lhi of channel l31 (+ 32 i) of position 2 wid + e: the swizzled index is rd0 ^ 2 ks
            const float* ms = stage_base + (rd0 ^ (unsigned)(2 * ks));
            const float* vs = ms + WD_OP;
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    af[e][i] = ms[(e * 64 + i * 32) * WD_ROW];
                    bf[e][i] = vs[(e * 64 + i * 32) * WD_ROW];
                }
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[e][i][j] = mfma32(af[e][i], bf[e][j], acc[e][i][j]);
                    }
            if (ks == WD_KT / 4 - 1 && !early) {  // uniform per wave
                __builtin_amdgcn_sched_barrier(0);
                produce();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
    // ---- publish the partial block: [sp][ob][xi][f][c] -----------------------------------------------------
    float* out = a.partials + ((size_t)sp * nob + ob) * (size_t)(16 * 64 * 64);
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    out[((2 * wid + e) * 64 + i * 32 + mfma_row(r, lane)) * 64 + j * 32 + l31] = acc[e][i][j][r];
}

// dw[f][c][3][3] += G^T (sum_sp partial[sp][ob][.][f][c]) G,  G^T = [1 .5 .5 0; 0 .5 -.5 0; 0 .5 .5 1].
// block = 16 positions x the 64 channels c of one f: a thread sums four consecutive c of its position over the splits in order
// (four independent chains per element, as before: the same bits) with 16-byte loads -- 16 lanes read 256 contiguous bytes of a
// split's slab where the one-float version read 64 -- then 64 threads finish the 4x4 -> 3x3 transform.
__global__ __launch_bounds__(256) void wino_dw_fused_finalize_kernel(const float* __restrict__ partials, int splits,
                                                                     int fblocks, int cblocks, int F, int C,
                                                                     float* __restrict__ dw) {
    __shared__ float u[16][65];
    const int xi = threadIdx.x >> 4, cq = threadIdx.x & 15;
    const int nob = fblocks * cblocks;
    const int fl = blockIdx.x & 63, ob = blockIdx.x >> 6;
    const float* p = partials + (size_t)ob * (16 * 64 * 64) + ((size_t)xi * 64 + fl) * 64 + cq * 4;
    const size_t stride = (size_t)nob * (16 * 64 * 64);
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
    auto ld = [&](int sp) { return *reinterpret_cast<const float4*>(p + (size_t)sp * stride); };
    auto add = [](float4& s, const float4& v) { s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; };
    int sp = 0;
    for (; sp + 3 < splits; sp += 4) {
        const float4 v0 = ld(sp), v1 = ld(sp + 1), v2 = ld(sp + 2), v3 = ld(sp + 3);
        add(s0, v0); add(s1, v1); add(s2, v2); add(s3, v3);
    }
    for (; sp < splits; ++sp) add(s0, ld(sp));
    u[xi][cq * 4 + 0] = (s0.x + s1.x) + (s2.x + s3.x);
    u[xi][cq * 4 + 1] = (s0.y + s1.y) + (s2.y + s3.y);
    u[xi][cq * 4 + 2] = (s0.z + s1.z) + (s2.z + s3.z);
    u[xi][cq * 4 + 3] = (s0.w + s1.w) + (s2.w + s3.w);
    __syncthreads();
    const int q = threadIdx.x;
    const int f = (ob % fblocks) * 64 + fl, c = (ob / fblocks) * 64 + q;
    if (q < 64 && f < F && c < C) {
        float m[4][4];
#pragma unroll
        for (int k = 0; k < 16; ++k) m[k >> 2][k & 3] = u[k][q];
        float t[3][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[0][j] = m[0][j] + 0.5f * (m[1][j] + m[2][j]);
            t[1][j] = 0.5f * (m[1][j] - m[2][j]);
            t[2][j] = 0.5f * (m[1][j] + m[2][j]) + m[3][j];
        }
        float* o = dw + ((size_t)f * C + c) * 9;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            o[r * 3 + 0] += t[r][0] + 0.5f * (t[r][1] + t[r][2]);
            o[r * 3 + 1] += 0.5f * (t[r][1] - t[r][2]);
            o[r * 3 + 2] += 0.5f * (t[r][1] + t[r][2]) + t[r][3];
        }
    }
}

struct WinoDwPlan {
    bool ok;
    int fblocks, cblocks, splits;
    unsigned T, tiles_per_split;
    size_t partial_floats;
};

static int g_wd_force = -1;  // experiment build: BCNN_HIP_WINOGRAD_DW_FUSED=0/1 overrides the rule
// cus: the CUs the launch may fill. On the side stream of a backward pass (conv.hip: conv_side_stream_deferred) a quarter of
// the chip is left to the pass's critical chain -- the sweeps and data gradients of the layers in front --, whose kernels
// cannot share a CU with this one's 128 KB workgroups: 192 of 256 measured best (224: +0.10 ms, 160: +0.15 ms per ResNet-18 step).
static WinoDwPlan wino_dw_fused_plan(const ConvShape& s, int cus = kCUs) {
    WinoDwPlan p;
    p.ok = false; p.partial_floats = 0;
    if (s.ksz != 3 || s.stride != 1 || s.pad != 1 || s.groups != 1) return p;
    const int TW = (s.W + 1) / 2, TH = (s.H + 1) / 2;
    if (TW < 4 || TH < 2 || s.C < 32 || s.F < 32) return p;  // the tile walk's carries; 16-byte row loads (W >= 7)
    if ((size_t)s.N * s.C * s.HW * 4 >= 0x7ffffff0ull || (size_t)s.N * s.F * s.HW * 4 >= 0x7ffffff0ull) return p;
    if (g_wd_force < 0) {
        const char* e = BCNN_EXP_ENV("BCNN_HIP_WINOGRAD_DW_FUSED");
        g_wd_force = e ? (e[0] == '0' ? 0 : 1) : 2;
    }
    if (g_wd_force == 0) return p;
    if (g_wd_force == 2 && (s.C < 64 || s.F < 64)) return p;
    p.T = (unsigned)((long long)s.N * TH * TW);
    p.fblocks = (s.F + 63) / 64; p.cblocks = (s.C + 63) / 64;
    const int nob = p.fblocks * p.cblocks;
    static const char* const dw_cus_str = BCNN_EXP_ENV("BCNN_HIP_DW_CUS");  // experiment override
    static const int dw_cus_env = dw_cus_str ? atoi(dw_cus_str) : 0;
    int splits = (dw_cus_env > 0 ? dw_cus_env : cus) / nob;  // one 128 KB workgroup per CU
    if (splits < 1) splits = 1;
    unsigned per = (p.T + (unsigned)splits - 1) / (unsigned)splits;
    per = (per + WD_KT - 1) / WD_KT * WD_KT;
    if (per < 4 * WD_KT) per = 4 * WD_KT;
    p.tiles_per_split = per;
    p.splits = (int)((p.T + per - 1) / per);
    if (g_wd_force == 2 && p.splits * nob < cus / 2) return p;  // too few tiles to fill the chip
    p.partial_floats = (size_t)p.splits * nob * (16 * 64 * 64);
    p.ok = true;
    return p;
}

size_t conv_dw_winograd_fused_workspace_floats(const ConvShape& s) { return wino_dw_fused_plan(s).partial_floats; }

bool conv_backward_weights_winograd_fused(const float* x, const float* dy, float* dw, const ConvShape& s, float* workspace,
                                          size_t workspace_floats) {
    // (the workspace was sized for the full plan.) Only on planes from 28 x 28 up: there the chain's sweeps are long enough to need
    // the CUs; with the 56 x 56 / 28 x 28 layers on conv_winograd43_dw.hip what is left here in ResNet-18 are the 14 x 14 / 7 x 7
    // layers, which do better on the whole chip (-0.05 ms per step)
    const bool yield_cus = conv_side_stream_deferred() && s.HW >= 784 && wino_dw_fused_plan(s).ok;
    const WinoDwPlan p = wino_dw_fused_plan(s, yield_cus ? kCUs * 3 / 4 : kCUs);
    if (!p.ok) return false;
    if (reinterpret_cast<uintptr_t>(workspace) & 15) return false;  // the finalize kernel reads the slabs 16 bytes at a time
    if (workspace == nullptr || workspace_floats < p.partial_floats) {
        fprintf(stderr, "[bcnn_hip] conv backward: workspace too small (%zu floats given, %zu needed)\n", workspace_floats,
                p.partial_floats);
        exit(1);
    }
    KTimer kt(K_CONV_DW_WINO, wf_flops(s), wf_bytes(s), wf_useful_flops(s));
    WinoDwArgs a;
    a.x = x; a.dy = dy; a.partials = workspace;
    a.N = s.N; a.C = s.C; a.F = s.F; a.H = s.H; a.W = s.W; a.TH = (s.H + 1) / 2; a.TW = (s.W + 1) / 2;
    a.T = p.T; a.tiles_per_split = p.tiles_per_split;
    a.fblocks = p.fblocks; a.cblocks = p.cblocks; a.splits = p.splits;
    a.x_bytes = (unsigned)((size_t)s.N * s.C * s.HW * 4);
    a.dy_bytes = (unsigned)((size_t)s.N * s.F * s.HW * 4);
    const int nob = p.fblocks * p.cblocks;
    trace_kernel("wino_dw_fused_kernel");
    if (s.W & 1) wino_dw_fused_kernel<true><<<(unsigned)(p.splits * nob), 512, 0, current_stream()>>>(a);
    else wino_dw_fused_kernel<false><<<(unsigned)(p.splits * nob), 512, 0, current_stream()>>>(a);
    KERNEL_CHECK();
    wino_dw_fused_finalize_kernel<<<(unsigned)(nob * 64), 256, 0, current_stream()>>>(workspace, p.splits, p.fblocks, p.cblocks,
                                                                                      s.F, s.C, dw);
    KERNEL_CHECK();
    return true;
}

}  // namespace bcnn_hip
