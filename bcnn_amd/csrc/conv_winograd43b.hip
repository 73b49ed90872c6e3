// conv_winograd43b.hip -- Winograd F(4x4, 3x3) forward / dX for planes of whole 4 x 4 tiles, second form (round 6).
//
// What the first form (conv_winograd43.hip: 32 channels x 32 tiles per unit on v_mfma_f32_32x32x2_f32, twelve waves) spent its
// time on besides MFMAs (profiles/r05_sq_step_resnet18.txt: matrix pipe 0.36 busy, waves 0.37 waiting): the 147 KB round trip of
// M through LDS for the output transform with four barriers per unit, an input transform that every 32-channel block repeats,
// and a serial section per chunk in which four of twelve waves transform while eight wait at the barrier. This form removes
// all three by choosing the MFMA whose RESULT layout is the output transform's input:
//
//   v_mfma_f32_16x16x4_f32: D[16 x 16] += A[16 x 4] B[4 x 16]; lane l holds A[m = l % 16][k = l / 16], B[k = l / 16][n = l % 16]
//   and D[m = 4 (l / 16) + i][n = l % 16], i = 0..3. With m = output channel and n = tile, a wave that owns ALL 36 positions of
//   a 16-channel x 16-tile block (36 accumulators x 4 registers = 144) ends the reduction with the 36 positions of (4 channels,
//   1 tile) in every lane: A^T M A happens in registers, no LDS, no barrier, and the 4 x 4 outputs leave as 16-byte stores whose
//   16 lanes of a DPP row are 16 consecutive tiles (256 contiguous bytes of an image row).
//
//   unit      = 64 output channels x 32 tiles x all 36 positions; 8 waves (two per SIMD, 256 registers each): wave (cb, tbw)
//               owns channels 16 cb .. + 15 and tiles 16 tbw .. + 15. V is transformed once per 64 output channels.
//   K loop    = "periods" of 8 reduction channels, each two sub-chunks of 4 (one MFMA step per position). LDS 147,456 B:
//               U stage [sub-chunk parity][xi/4 9][k 4][channel 64][xi%4 4]   2 x 36,864 B, by LDS-DMA (the pack kernel writes U
//                                                                             in this very order: a stage is 36 linear KB pieces)
//               V stage [period parity][c 2][xi/4 9][k 4][tile 32][xi%4 4]    2 x 36,864 B, written by the transforming waves
//               so that the A and B fragments of FOUR positions are one conflict-free ds_read_b128 each: 18 LDS reads per 36 MFMAs.
//   roles     = the waves form two groups (waves 0-3 / 4-7: one wave of each per SIMD) that alternate by period. In period r the
//               group T = (r + 1) & 1 transforms the NEXT period's V at the start of the first sub-chunk (one patch per lane,
//               from registers requested two periods earlier) and issues both sub-chunks' U requests; the group R = r & 1
//               requests the patches of period r + 2. A wave therefore never waits for a U piece while patch requests of its
//               own are in flight (one in-order counter per wave), the patches have two periods to arrive, and the transform of a
//               wave runs while its SIMD partner multiplies -- its memory and LDS stalls cost nothing.
//   one barrier per sub-chunk (36 MFMAs per wave); none in the epilogue. The stream of periods runs across units: the next
//   unit's first V and U are in place when the current unit's epilogue ends.
// K-split tail as in the first form (the units of a last, partly filled round as period ranges dealt out over all CUs, raw
// partial outputs to scratch, wino43b_tail_fixup_kernel adds them in channel order).
//
// Reference: bcnn_forward_conv_layer_cpu / bcnn_backward_conv_layer_cpu (bcnn_conv_layer.c:438-481, 533-581) are what is
// computed; the reference's own transformed-domain path (bcnn_conv_layer.c:388-436 on bcnn_mat.c:1403-2138) is the precedent.
#include "conv_common.h"
#include "lds_dma.h"
#include "wino43_math.h"
#include "wino43_pack.h"
#include "wino43b_mma.h"

namespace bcnn_hip {

struct Wino43bArgs {
    const float* src;  // x (forward) or dy (dX): [N][J][H][W]
    const float* upk;  // transformed weights in stage order (wino43_pack_one, layout 1), zero padded
    float* dst;        // [N][M][H][W]
    float* stats;      // optional: [M][2 tblocks][2]
    int N, J, M, H, W, TH, TW;
    unsigned T;
    int nper, mblocks, tblocks, nunits;
    unsigned src_bytes, upk_bytes, dst_bytes, stats_bytes;
    // K-split tail (see Wino43Args): units [nunits, nunits + tail_units) are run as period ranges; workgroup i takes periods
    // [i * tail_q, (i + 1) * tail_q) of the flattened tail as one or two pieces -> tail_scr[2 i + piece][channel 64][tile 32][16]
    float* tail_scr;
    int tail_q, tail_units;
    unsigned tail_scr_bytes;
};

// A period of the workgroup's stream: item `it` (a unit or a tail piece), period kp of its np, packed into four scalar
// registers (three of these live across the whole loop; scalar registers are what this kernel runs out of first).
struct WbCur {
    int it;    // -1: past the end of the stream
    int kpnp;  // kp | np << 16
    int kbmb;  // kb | mb << 16  (first period of the item within its unit; channel block)
    int tb;    // tile block
    __device__ __forceinline__ bool valid() const { return it >= 0; }
    __device__ __forceinline__ int kp() const { return kpnp & 0xffff; }
    __device__ __forceinline__ int np() const { return kpnp >> 16; }
    __device__ __forceinline__ int kb() const { return kbmb & 0xffff; }
    __device__ __forceinline__ int mb() const { return kbmb >> 16; }
    __device__ __forceinline__ int period() const { return kb() + kp(); }  // of the unit
};

template <bool STATS>
__global__ __launch_bounds__(64 * WB_NW, 2) void wino43b_kernel(const Wino43bArgs a) {
    __shared__ __attribute__((aligned(1024))) float lds[2 * WB_USTAGE + 2 * WB_VSTAGE];  // 147,456 bytes
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid >> 2, tw4 = wid & 3;   // role group; transform role: channels 2 tw4, 2 tw4 + 1 of a period
    const int cb = wid & 3, tbw = wid >> 2;    // MFMA role: channels 16 cb .., tiles 16 tbw ..
    const int l31 = lane & 31, lhi = lane >> 5;
    const int l15 = lane & 15, lq = lane >> 4;
    const int HW = a.H * a.W;
    const int bid = (int)blockIdx.x, grid = (int)gridDim.x;
    const rsrc_i4 rs_src = make_rsrc(a.src, a.src_bytes);
    const rsrc_i4 rs_u = make_rsrc(a.upk, a.upk_bytes);
    const rsrc_i4 rs_dst = make_rsrc(a.dst, a.dst_bytes);
    const unsigned lds0 = lds_offset(&lds[0]);
    const unsigned per_img = (unsigned)(a.TH * a.TW);
    const unsigned row_bytes = (unsigned)a.W * 4u;
    const int nc4 = 2 * a.nper;  // sub-chunks of 4 reduction channels in the packed weights

    // ---- this workgroup's items: whole units, then (K-split tail) one or two pieces ----
    const int nreg = bid < a.nunits ? (a.nunits - 1 - bid) / grid + 1 : 0;
    int npieces = 0, pa_unit = 0, pa_k0 = 0, pa_n = 0, pb_n = 0;
    if (a.tail_units > 0) {
        const int c0 = bid * a.tail_q, c1 = min(c0 + a.tail_q, a.tail_units * a.nper);
        if (c0 < c1) {
            pa_unit = a.nunits + c0 / a.nper;
            pa_k0 = c0 % a.nper;
            pa_n = min(c1 - c0, a.nper - pa_k0);
            pb_n = (c1 - c0) - pa_n;
            npieces = pb_n > 0 ? 2 : 1;
        }
    }
    const int nitems = nreg + npieces;
    if (nitems == 0) return;
    auto load_item = [&](int it, WbCur& c) {
        int unit, kb, np;
        if (it < nreg) { unit = bid + it * grid; kb = 0; np = a.nper; }
        else if (it == nreg) { unit = pa_unit; kb = pa_k0; np = pa_n; }
        else { unit = pa_unit + 1; kb = 0; np = pb_n; }
        const int tb = unit / a.mblocks;  // channel blocks of one tile block run together
        c.it = it; c.kpnp = np << 16; c.kbmb = kb | ((unit - tb * a.mblocks) << 16); c.tb = tb;
    };
    auto advance = [&](WbCur& c) {
        if (!c.valid()) return;
        if (c.kp() + 1 < c.np()) { ++c.kpnp; return; }
        if (c.it + 1 < nitems) { load_item(c.it + 1, c); return; }
        c.it = -1;
    };
    auto slot_of = [&](const WbCur& c) { return c.it < nreg ? -1 : 2 * bid + (c.it - nreg); };

    // ---- patch requests of the transforming role: lane = (channel 2 tw4 + lhi of the period, tile l31 of the tile block) ----
    // The two columns either side of a lane's own four are the neighbouring lanes' (DPP wave shifts) except at the ends of a
    // half-wave: lane l31 = 0 lacks its left column, l31 = 31 its right one. Those 2 x 6 values are fetched by lanes that have
    // nothing else to do in that instruction -- lanes l31 = 1..6 rows 0..5 of lane 0's left column, lanes 25..30 those of lane
    // 31's right column: ONE dword request per patch instead of six almost empty ones -- and handed over with ds_bpermute when
    // the patch is transformed.
    unsigned v_mid_ = kOOB;  // byte offset of patch row 1 (image row 4 th), own columns, channel lhi of the wave's pair
    unsigned e_off_ = kOOB;  // byte offset of this lane's edge duty (out of range: none, or that value is padding)
    unsigned pflags = 0;     // 1 row 0 exists, 2 row 5 exists, 16 / 32 left / right padding
    int dec_tb = -1;         // the tile block those belong to (this group's last request)
    auto tile_coords = [&](unsigned t, unsigned& n, int& th, int& tw) {
        n = t / per_img;
        const unsigned rr = t - n * per_img;
        th = (int)(rr / (unsigned)a.TW);
        tw = (int)(rr - (unsigned)th * (unsigned)a.TW);
    };
    auto decode = [&](int tb) {
        dec_tb = tb;
        const unsigned t = (unsigned)tb * WB_BT + (unsigned)l31;
        const bool ok = t < a.T;
        unsigned n = 0; int th = 0, tw = 0;
        if (ok) tile_coords(t, n, th, tw);
        const unsigned chan = (unsigned)lhi * (unsigned)HW * 4u;
        v_mid_ = ok ? n * (unsigned)a.J * (unsigned)HW * 4u + chan + (unsigned)(4 * th * a.W + 4 * tw) * 4u : kOOB;
        // rows 1..4 of a whole tile always exist; rows 0 and 5 are padding at the top / bottom tile row
        pflags = !ok ? 0u : (th > 0 ? 1u : 0u) | (4 * th + 4 < a.H ? 2u : 0u) | (tw == 0 ? 16u : 0u) | (tw + 1 == a.TW ? 32u : 0u);
        // edge duty: row (l31 - 1) of the first tile's left column / row (l31 - 25) of the last tile's right column
        const bool left = l31 >= 1 && l31 <= 6, right = l31 >= 25 && l31 <= 30;
        const unsigned td = (unsigned)tb * WB_BT + (left ? 0u : 31u);
        const int row = left ? l31 - 1 : l31 - 25;
        e_off_ = kOOB;
        if ((left || right) && td < a.T) {
            unsigned dn; int dth, dtw;
            tile_coords(td, dn, dth, dtw);
            const int ih = 4 * dth - 1 + row, iw = left ? 4 * dtw - 1 : 4 * dtw + 4;
            if (ih >= 0 && ih < a.H && iw >= 0 && iw < a.W)
                e_off_ = dn * (unsigned)a.J * (unsigned)HW * 4u + chan + (unsigned)(ih * a.W + iw) * 4u;
        }
    };
    // a patch = p[6]: own columns (kept as the 16-byte tuples the requests fill) + eduty: this lane's edge duty.
    // live = false: the same seven instructions with every lane out of range (nothing is fetched, zeros come back).
    auto load_patch = [&](const WbCur& c, buf_f32x4 (&p)[6], float& eduty, bool live) {
        if (live && c.tb != dec_tb) decode(c.tb);
        // The row step rides in the scalar offset (the range check sees the vector offset only); out-of-range offsets return 0.0
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((c.period() * WB_KP + 2 * tw4) * HW * 4);
        const unsigned v_mid = live ? v_mid_ : kOOB, e_off = live ? e_off_ : kOOB;
        const unsigned v_top = ((pflags & 1u) && live) ? v_mid - row_bytes : kOOB, v_bot = (pflags & 2u) ? v_mid : kOOB;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const unsigned so = i == 0 ? soff : (unsigned)__builtin_amdgcn_readfirstlane((int)(soff + (unsigned)(i - 1) * row_bytes));
            p[i] = buffer_load_f32x4(rs_src, (int)(i == 0 ? v_top : i == 5 ? v_bot : v_mid), (int)so, 0);
        }
        eduty = buffer_load_f32(rs_src, (int)e_off, (int)soff, 0);
    };
    // B^T d B -> V[vs][c = tw4 / 2][xi / 4][k = 2 (tw4 % 2) + lhi][tile l31][xi % 4]: nine 16-byte writes, a wave's 64 lanes 1 KB each
    auto write_v = [&](int vs, const buf_f32x4 (&p)[6], const float eduty) {
        const bool pad_l = (pflags & 16u) != 0, pad_r = (pflags & 32u) != 0;
        // the edge values: lane l31 = 0 takes row i from lane + 1 + i, lane l31 = 31 from lane - 6 + i (others: unused)
        const int esrc = 4 * (l31 == 0 ? lane + 1 : lane - 6);
        float tt[6][6];  // columns first: tt[.][j] = B^T d[.][j]
        {   // the two neighbour columns first: they read the neighbouring lanes' own columns
            float cl[6], cr[6], out[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int ev = __builtin_amdgcn_ds_bpermute(esrc + 4 * i, __float_as_int(eduty));
                // (through scalars: hipcc 7.2 reads element 0 for `__builtin_bit_cast(int, vector[3])`)
                const float own_first = p[i][0], own_last = p[i][3];
                // lane l - 1's last own column; the first lane of a half-wave takes the fetched value
                const int l = __builtin_amdgcn_update_dpp(ev, __float_as_int(own_last), 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                cl[i] = pad_l ? 0.f : __builtin_bit_cast(float, l31 == 0 ? ev : l);
                // lane l + 1's first own column
                const int r = __builtin_amdgcn_update_dpp(ev, __float_as_int(own_first), 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
                cr[i] = pad_r ? 0.f : __builtin_bit_cast(float, l31 == 31 ? ev : r);
            }
            w43_bt(cl, out);
#pragma unroll
            for (int i = 0; i < 6; ++i) tt[i][0] = out[i];
            w43_bt(cr, out);
#pragma unroll
            for (int i = 0; i < 6; ++i) tt[i][5] = out[i];
        }
#pragma unroll
        for (int j = 1; j <= 4; ++j) {
            const float col[6] = {p[0][j - 1], p[1][j - 1], p[2][j - 1], p[3][j - 1], p[4][j - 1], p[5][j - 1]};
            float out[6];
            w43_bt(col, out);
#pragma unroll
            for (int i = 0; i < 6; ++i) tt[i][j] = out[i];
        }
        float* v = lds + 2 * WB_USTAGE + vs * WB_VSTAGE + (tw4 >> 1) * WB_VC + ((2 * (tw4 & 1) + lhi) * 32 + l31) * 4;
#pragma unroll
        for (int ip = 0; ip < 3; ++ip) {  // two rows = twelve positions = three writes at a time
            float o0[6], o1[6];
            w43_bt(tt[2 * ip], o0);
            w43_bt(tt[2 * ip + 1], o1);
            *reinterpret_cast<f32x4*>(v + (3 * ip) * 512) = f32x4{o0[0], o0[1], o0[2], o0[3]};
            *reinterpret_cast<f32x4*>(v + (3 * ip + 1) * 512) = f32x4{o0[4], o0[5], o1[0], o1[1]};
            *reinterpret_cast<f32x4*>(v + (3 * ip + 2) * 512) = f32x4{o1[2], o1[3], o1[4], o1[5]};
        }
    };
    // one KB piece of a U sub-chunk: the packed weights hold every (channel block, sub-chunk) stage as 36 linear pieces
    auto dma_u = [&](const WbCur& c, int sub, int piece) {
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(((c.mb() * nc4 + 2 * c.period() + sub) * 36 + piece) * 1024);
        dma_row_x4(rs_u, lds0 + (unsigned)((sub * WB_USTAGE) * 4 + piece * 1024), (unsigned)lane * 16u, soff);
    };

    f32x4 acc[36];
    const float* const ufrag = lds + (lq * 64 + 16 * cb + l15) * 4;                      // + sub * WB_USTAGE + (xi / 4) * 1024
    const float* const vfrag = lds + 2 * WB_USTAGE + (lq * 32 + 16 * tbw + l15) * 4;    // + vs * WB_VSTAGE + c * WB_VC + (xi / 4) * 512

    // A^T M A on the accumulators, stores and statistics of item c (the wave's 16 channels x 16 tiles)
    auto epilogue = [&](const WbCur& c) {
        const unsigned t = (unsigned)c.tb * WB_BT + (unsigned)(16 * tbw + l15);
        const bool ok = t < a.T;
        const unsigned n = ok ? t / per_img : 0u;
        const unsigned rr = ok ? t - n * per_img : 0u;
        const int th = (int)(rr / (unsigned)a.TW);
        const int tw = (int)(rr - (unsigned)th * (unsigned)a.TW);
        const int fl0 = 16 * cb + 4 * lq;           // first of this lane's four channels within the unit
        const int f0 = c.mb() * WB_BF + fl0;
        const int slot = slot_of(c);
        const rsrc_i4 rs_stats = make_rsrc(a.stats, STATS ? a.stats_bytes : 0u);
        const rsrc_i4 rs_scr = make_rsrc(a.tail_scr, a.tail_scr_bytes);
        const unsigned o00 = ok ? (n * (unsigned)a.M * (unsigned)HW + (unsigned)(4 * th * a.W + 4 * tw)) * 4u : kOOB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float tt[6][4];  // rows first: tt[r][.] = A^T applied along the columns of row r
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const float m[6] = {acc[6 * r][i], acc[6 * r + 1][i], acc[6 * r + 2][i], acc[6 * r + 3][i], acc[6 * r + 4][i], acc[6 * r + 5][i]};
                float y[4];
                w43_at(m, y);
#pragma unroll
                for (int q = 0; q < 4; ++q) tt[r][q] = y[q];
            }
            float o[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float m[6] = {tt[0][q], tt[1][q], tt[2][q], tt[3][q], tt[4][q], tt[5][q]};
                float y[4];
                w43_at(m, y);
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r][q] = y[r];
            }
            if (slot >= 0) {  // uniform: a piece of a K-split unit -- raw partial outputs to its scratch slot
                const unsigned so = (unsigned)(((slot * WB_BF + fl0 + i) * WB_BT + 16 * tbw + l15) * 64);
#pragma unroll
                for (int r = 0; r < 4; ++r) wb_store_x4(f32x4{o[r][0], o[r][1], o[r][2], o[r][3]}, rs_scr, so + 16u * r, 0u);
            } else {
                const int f = f0 + i;
                const bool f_ok = f < a.M;
                const unsigned off = (f_ok && ok) ? o00 + (unsigned)f * (unsigned)HW * 4u : kOOB;
#pragma unroll
                for (int r = 0; r < 4; ++r) wb_store_x4(f32x4{o[r][0], o[r][1], o[r][2], o[r][3]}, rs_dst, off, (unsigned)r * row_bytes);
                if (STATS) {  // the 16 lanes of a DPP row hold channel f for the wave's 16 tiles
                    float sv = 0.f, sq = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int q = 0; q < 4; ++q) { sv += o[r][q]; sq += o[r][q] * o[r][q]; }
                    if (!ok) { sv = 0.f; sq = 0.f; }
                    sv = row16_sum_dpp(sv);
                    sq = row16_sum_dpp(sq);
                    const unsigned so = (l15 == 0 && f_ok) ? (unsigned)((f * 2 * a.tblocks + 2 * c.tb + tbw) * 8) : kOOB;
                    buffer_store_f32x2(buf_f32x2{sv, sq}, rs_stats, (int)so, 0, 0);
                }
            }
        }
    };

    // ---- prologue: period 0's V (group 0) and first U sub-chunk into stage 0; period 1's patches into group 1's registers ----
    WbCur c0, c1, c2;
    load_item(0, c0);
    c1 = c0; advance(c1);
    c2 = c1; advance(c2);
    for (int q = wid; q < 36; q += WB_NW) dma_u(c0, 0, q);  // older than the patch requests below
    buf_f32x4 p[6]; float eduty;
    if (grp == 0) {
        load_patch(c0, p, eduty, true);
        dma_wait();
        write_v(0, p, eduty);
    }
    load_patch(c1, p, eduty, grp == 1 && c1.valid());  // (blanks in group 0)
    dma_wait_n<7>();                                   // the U pieces have landed; the seven patch requests fly on

    // The stream of periods. Roles alternate per group and period: in period gp the group T = (gp + 1) & 1 transforms the NEXT
    // period's V -- after its MFMAs of the first sub-chunk, from the patches it requested a period ago in the other role -- and
    // requests that period's U stage 0 in the second sub-chunk; the group R requests this period's U stage 1 right behind
    // the first barrier and, after its MFMAs, the patches of period gp + 2 (its U pieces are OLDER than its patch requests, so
    // `vmcnt(7)` waits for exactly them: one in-order counter per wave). What the order of these blocks owes to hipcc: the patch
    // registers are (re)defined at ONE place of the loop by both roles (T: blank requests -- every lane out of range, nothing
    // fetched -- behind its use), so they reach the back edge as one value; requested under one branch and used under another,
    // or defined as zeros on the other path, they are merged with copies behind `s_waitcnt vmcnt(0)` and the request loses
    // its lead.
    for (int gp = 0; c0.valid(); ++gp) {
        const bool t_role = grp == ((gp + 1) & 1);
        const int vs = gp & 1;
        // ---- first sub-chunk: U stage 0, V[vs][0] ----
        lds_barrier();  // U stage 0 and V stage vs are complete; the other stages' readers are done
        if (!t_role) {
#pragma unroll
            for (int q = 0; q < 9; ++q) dma_u(c0, 1, 9 * tw4 + q);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (c0.kp() == 0) wb_mma<true>(acc, ufrag, vfrag + vs * WB_VSTAGE);
        else wb_mma<false>(acc, ufrag, vfrag + vs * WB_VSTAGE);
        __builtin_amdgcn_sched_barrier(0);
        if (t_role && c1.valid()) {
            dma_wait();  // the patches (and whatever an epilogue stored since)
            write_v(vs ^ 1, p, eduty);
        }
        load_patch(c2, p, eduty, !t_role && c2.valid());
        // ---- second sub-chunk: U stage 1, V[vs][1] ----
        if (!t_role) dma_wait_n<7>();  // its pieces of U stage 1 (older than the seven requests)
        lds_barrier();
        if (t_role && c1.valid()) {
#pragma unroll
            for (int q = 0; q < 9; ++q) dma_u(c1, 0, 9 * tw4 + q);
        }
        __builtin_amdgcn_sched_barrier(0);
        wb_mma<false>(acc, ufrag + WB_USTAGE, vfrag + vs * WB_VSTAGE + WB_VC);
        __builtin_amdgcn_sched_barrier(0);
        // its pieces of the next period's U stage 0 have had 36 MFMAs to land; waiting here, before an epilogue puts stores
        // into the same counter, keeps the next barrier free of memory waits
        if (t_role) dma_wait();
        if (c0.kp() + 1 == c0.np()) epilogue(c0);
        c0 = c1; c1 = c2; advance(c2);
    }
}

// K-split tail of wino43b_kernel: the pieces of tail unit u (unit index nunits + u) sit in the scratch slots of the workgroups
// whose period ranges met it -- workgroup i's range starts at period i * tail_q of the flattened tail; a range that started in
// the previous unit left its SECOND piece here (slot 2 i + 1), every other one its first (slot 2 i). They are added in channel
// order (ascending i), stored, and counted into the batch-norm statistics (slot 2 tb of the unit's two; the other gets zeros).
// One wave per (unit, channel): lane = (tile, upper / lower two rows of its 4 x 4 outputs).
template <bool STATS>
__global__ __launch_bounds__(256) void wino43b_tail_fixup_kernel(const Wino43bArgs a) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int u = blockIdx.x >> 4, fl = (blockIdx.x & 15) * 4 + wid;
    const int NC = a.nper;
    const int unit = a.nunits + u;
    const int mb = unit % a.mblocks, tb = unit / a.mblocks;
    const int f = mb * WB_BF + fl;
    const bool f_ok = f < a.M;
    const int tile = lane >> 1, half = lane & 1;
    const unsigned t = (unsigned)tb * WB_BT + (unsigned)tile;
    const bool tile_ok = t < a.T;
    const unsigned per_img = (unsigned)(a.TH * a.TW);
    const unsigned n = tile_ok ? t / per_img : 0u;
    const unsigned rr = tile_ok ? t - n * per_img : 0u;
    const int th = (int)(rr / (unsigned)a.TW), tw = (int)(rr - (unsigned)th * (unsigned)a.TW);
    const int c0 = u * NC, c1 = c0 + NC;
    const int i0 = c0 / a.tail_q, i1 = (c1 - 1) / a.tail_q;
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0;
    for (int i = i0; i <= i1; ++i) {
        const int slot = 2 * i + (i * a.tail_q < c0 ? 1 : 0);
        const float4* v = reinterpret_cast<const float4*>(a.tail_scr + ((size_t)(slot * WB_BF + fl) * WB_BT + tile) * 16 + half * 8);
        const float4 x0 = v[0], x1 = v[1];
        r0.x += x0.x; r0.y += x0.y; r0.z += x0.z; r0.w += x0.w;
        r1.x += x1.x; r1.y += x1.y; r1.z += x1.z; r1.w += x1.w;
    }
    const int HW = a.H * a.W;
    const int row0 = 4 * th + 2 * half;
    if (f_ok && tile_ok) {
        float* d = a.dst + ((size_t)n * a.M + f) * HW + (size_t)row0 * a.W + 4 * tw;
        *reinterpret_cast<float4*>(d) = r0;
        *reinterpret_cast<float4*>(d + a.W) = r1;
    }
    if (STATS) {
        float sv = 0.f, sq = 0.f;
        if (tile_ok) {
            sv = r0.x + r0.y + r0.z + r0.w + r1.x + r1.y + r1.z + r1.w;
            sq = r0.x * r0.x + r0.y * r0.y + r0.z * r0.z + r0.w * r0.w + r1.x * r1.x + r1.y * r1.y + r1.z * r1.z + r1.w * r1.w;
        }
        sv = wave_sum_dpp(sv);
        sq = wave_sum_dpp(sq);
        if (lane == 63 && f_ok) {
            float* st = a.stats + ((size_t)f * 2 * a.tblocks + 2 * tb) * 2;
            st[0] = sv; st[1] = sq; st[2] = 0.f; st[3] = 0.f;
        }
    }
}

__global__ __launch_bounds__(256) void wino43b_pack_weights_kernel(const float* __restrict__ w, float* __restrict__ u, int F, int C,
                                                                   int dx_mode, int Jpad, int Mpad) {
    wino43_pack_one(w, u, F, C, dx_mode, Jpad, Mpad, blockIdx.x * 256 + threadIdx.x, /*layout=*/1);
}

// ---- host side ------------------------------------------------------------------------------------------
struct WbScratch {
    float* p = nullptr;
    size_t cap = 0;
};
static float* wb_grow(WbScratch (&tab)[64], size_t floats, size_t min_floats) {
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) { fprintf(stderr, "[bcnn_hip] device ordinal %d out of range\n", dev); exit(1); }
    WbScratch& sc = tab[dev];
    if (sc.p == nullptr || sc.cap < floats) {
        if (sc.p) {
            HIP_CHECK(hipStreamSynchronize(current_stream()));
            HIP_CHECK(hipFree(sc.p));
        }
        const size_t cap = floats < min_floats ? min_floats : floats;
        HIP_CHECK(hipMalloc((void**)&sc.p, cap * sizeof(float)));
        sc.cap = cap;
    }
    return sc.p;
}
static thread_local WbScratch g_wb_u_scratch[64];
static thread_local WbScratch g_wb_tail_scratch[64];  // separate from the U scratch, which the running kernel reads

int wino43b_stats_slots(const ConvShape& s) {  // per channel, for a caller's statistics buffer
    const long long T = (long long)s.N * (s.H / 4) * (s.W / 4);
    return 2 * (int)ceil_div(T, WB_BT);
}

void wino43b_pack_dims(int J, int M, int* Jpad, int* Mpad) {
    *Jpad = (J + WB_KP - 1) / WB_KP * WB_KP;
    *Mpad = (M + WB_BF - 1) / WB_BF * WB_BF;
}

// src / dst: 16-byte aligned, whole 4 x 4 tiles (the caller checked: wino43_usable)
void wino43b_run(const float* src, const float* w, float* dst, const ConvShape& s, int dx_mode, ConvStats* stats) {
    Wino43bArgs a;
    a.src = src; a.dst = dst;
    a.N = s.N; a.J = dx_mode ? s.F : s.C; a.M = dx_mode ? s.C : s.F; a.H = s.H; a.W = s.W;
    a.TH = s.H / 4; a.TW = s.W / 4;
    a.T = (unsigned)((long long)s.N * a.TH * a.TW);
    int Jpad, Mpad;
    wino43b_pack_dims(a.J, a.M, &Jpad, &Mpad);
    a.nper = Jpad / WB_KP;
    a.mblocks = Mpad / WB_BF;
    a.tblocks = (int)((a.T + WB_BT - 1) / WB_BT);
    a.nunits = a.mblocks * a.tblocks;
    a.stats = (stats && stats->partials) ? stats->partials : nullptr;
    a.stats_bytes = a.stats ? (unsigned)((size_t)a.M * 2 * a.tblocks * 2 * sizeof(float)) : 0u;
    a.src_bytes = (unsigned)((size_t)s.N * a.J * s.HW * 4);
    a.dst_bytes = (unsigned)((size_t)s.N * a.M * s.HW * 4);
    const size_t u_floats = (size_t)36 * Jpad * Mpad;
    a.upk_bytes = (unsigned)(u_floats * 4);
    float* U = prepack_take(w, PREPACK_WINO, dx_mode, u_floats);  // transformed ahead by bcnn_hip_conv_prepack?
    if (!U) {
        U = wb_grow(g_wb_u_scratch, u_floats, (size_t)1 << 20);
        wino43b_pack_weights_kernel<<<ceil_div((long long)Jpad * Mpad, 256), 256, 0, current_stream()>>>(w, U, s.F, s.C, dx_mode, Jpad,
                                                                                                           Mpad);
        KERNEL_CHECK();
    }
    a.upk = U;
    const int nblocks = a.nunits;
    const unsigned grid = (unsigned)(nblocks < kCUs ? nblocks : kCUs);  // persistent: one 144 KB workgroup per CU
    // K-split tail: the blocks of a last, partly filled round as period ranges dealt out evenly over all CUs. Priced in period
    // times: a unit is nper periods + an epilogue (~1), a piece its periods + an epilogue, the fix-up launch ~2.
    a.tail_scr = nullptr; a.tail_q = 0; a.tail_units = 0; a.tail_scr_bytes = 0;
    static const int ksplit_on = BCNN_EXP_ENV("BCNN_HIP_NO_WINO_KSPLIT") ? 0 : 1;  // A/B switch of the experiment build
    const int rem = nblocks % (int)grid;
    if (ksplit_on && rem > 0 && nblocks > (int)grid) {
        const int NC = a.nper;
        const int q = (int)ceil_div((long long)rem * NC, (long long)grid);
        const double ep = 1.0;
        if (q >= 1 && q <= NC && q + 2 * ep + 2.0 < NC + ep) {
            a.nunits = nblocks - rem;
            a.tail_units = rem;
            a.tail_q = q;
            const size_t scr_floats = (size_t)2 * grid * WB_BF * WB_BT * 16;
            a.tail_scr = wb_grow(g_wb_tail_scratch, scr_floats, scr_floats);
            a.tail_scr_bytes = (unsigned)(scr_floats * sizeof(float));
        }
    }
    trace_kernel(dx_mode ? "wino43b_kernel:dx" : "wino43b_kernel:fwd");
    if (a.stats) wino43b_kernel<true><<<grid, 64 * WB_NW, 0, current_stream()>>>(a);
    else wino43b_kernel<false><<<grid, 64 * WB_NW, 0, current_stream()>>>(a);
    KERNEL_CHECK();
    if (a.tail_units > 0) {
        trace_kernel("wino43b_tail_fixup");
        if (a.stats) wino43b_tail_fixup_kernel<true><<<(unsigned)(a.tail_units * 16), 256, 0, current_stream()>>>(a);
        else wino43b_tail_fixup_kernel<false><<<(unsigned)(a.tail_units * 16), 256, 0, current_stream()>>>(a);
        KERNEL_CHECK();
    }
    if (stats) stats->splits = a.stats ? 2 * a.tblocks : 0;
}

}  // namespace bcnn_hip
